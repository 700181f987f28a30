// valu_issue_microbench.hip — how fast does gfx950 issue the integer VALU instructions k_fast_cells is made of?
// Register-only chains (8 independent accumulators per lane, one inline-asm instruction each per trip), 256 CUs x 8 waves/SIMD resident, no memory traffic.
// Prints wave-instructions per second per instruction class; build: hipcc --offload-arch=gfx950 -O3 tools/valu_issue_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef short pk2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096, kAcc = 8;

#define ORBX_OP(txt) asm volatile(txt : "+v"(a[i]) : "v"(b), "v"(c))
template <int OP> __global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
    uint32_t a[kAcc], b = seed ^ threadIdx.x, c = seed * 2654435761u + blockIdx.x;
#pragma unroll
    for (int i = 0; i < kAcc; i++) a[i] = b * (i + 1) + c;
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < kAcc; i++) {              // inline asm: exactly one instruction of the class per accumulator and trip
            if (OP == 0) ORBX_OP("v_min_i32 %0, %0, %1");
            if (OP == 1) ORBX_OP("v_pk_min_i16 %0, %0, %1");
            if (OP == 2) ORBX_OP("v_pk_max_i16 %0, %0, %2");
            if (OP == 3) ORBX_OP("v_perm_b32 %0, %0, %1, %2");
            if (OP == 4) ORBX_OP("v_alignbyte_b32 %0, %0, %1, 1");
            if (OP == 5) ORBX_OP("v_pk_mad_i16 %0, %0, %1, %2");
            if (OP == 6) ORBX_OP("v_add_u32 %0, %0, %1");
            if (OP == 7) ORBX_OP("v_dot4_u32_u8 %0, %0, %1, %2");
            if (OP == 8) ORBX_OP("v_pk_sub_i16 %0, %0, %1");
            if (OP == 9) ORBX_OP("v_fma_f32 %0, %0, %1, %2");
            if (OP == 10) ORBX_OP("v_pk_maximum3_f16 %0, %0, %1, %2");
            if (OP == 11) ORBX_OP("v_pk_min_f16 %0, %0, %1");
            if (OP == 12) ORBX_OP("v_max3_i16 %0, %0, %1, %2");
            if (OP == 13) ORBX_OP("v_pk_add_u16 %0, %0, %1");
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < kAcc; i++) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP> void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;                       // 8 workgroups of 4 waves per CU
    k<OP><<<blocks, 256>>>(d, 1);
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * kIters * kAcc;                  // wave-instructions of the class (loop overhead is scalar)
    const double rate = instr / (ms * 1e-3);
    printf("%-18s %8.3f ms  %7.1f G wave-instr/s  = %.2f clk per wave64 instruction per SIMD (1024 SIMDs @ 2.4 GHz)\n", name, ms, rate * 1e-9, 1024 * 2.4e9 / rate);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_min_i32", d); run<1>("v_pk_min_i16", d); run<2>("v_pk_max_i16", d); run<3>("v_perm_b32", d); run<4>("v_alignbyte_b32", d);
    run<5>("v_pk_mad_i16", d); run<6>("v_add_u32", d); run<7>("v_dot4_u32_u8", d); run<8>("v_pk_sub_i16", d); run<9>("v_fma_f32", d); run<10>("v_pk_maximum3_f16", d); run<11>("v_pk_min_f16", d); run<12>("v_max3_i16", d); run<13>("v_pk_add_u16", d);
    return 0;
}
