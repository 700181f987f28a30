// LDS byte-read microbenchmark: how do ds_read_u8 address patterns conflict on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int kIters = 2048;
template <int PAT, int WIDE> __global__ void __launch_bounds__(256) k(uint32_t* out) {
    __shared__ uint8_t lds[16384];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) lds[i] = (uint8_t)(i * 7);
    __syncthreads();
    int a;
    if (PAT == 0) a = lane;                                   // consecutive bytes
    if (PAT == 1) a = 4 * lane;                               // consecutive dwords
    if (PAT == 2) a = 48 * (lane / 10) + 4 * (lane % 10);     // phase-A like (rows of 10 dwords, pitch 48)
    if (PAT == 3) a = (int)((lane * 2654435761u) >> 19) & 4095;   // pseudo-random bytes in 4 KB
    if (PAT == 4) a = 2 * lane;
    if (PAT == 5) a = 48 * (lane & 31) + (lane >> 5);         // column walk: pitch 48
    if (PAT == 6) a = ((int)((lane * 2654435761u) >> 19) & 1023) * 4;  // random dwords
    a += (tid >> 6) * 4096;
    uint32_t acc = 0;
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int ad = (a + u * 52 + it) & 16383;
            if (WIDE) acc += *(const volatile uint32_t*)(lds + (ad & ~3)); else acc += *(const volatile uint8_t*)(lds + ad);
        }
    }
    out[blockIdx.x * 256 + tid] = acc;
}
template <int PAT, int WIDE> void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 4;
    k<PAT, WIDE><<<blocks, 256>>>(d);
    (void)hipEventRecord(e0); k<PAT, WIDE><<<blocks, 256>>>(d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * kIters * 16;     // wave-level LDS instructions
    printf("%-34s %8.3f ms  %.2f clk per wave LDS instruction per CU (256 CUs @ 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 256 / instr);
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0, 0>("u8 consecutive bytes", d); run<1, 0>("u8 stride 4", d); run<2, 0>("u8 rows of 10 dwords pitch 48", d); run<3, 0>("u8 random bytes", d);
    run<4, 0>("u8 stride 2", d); run<5, 0>("u8 column walk pitch 48", d); run<6, 0>("u8 random dwords", d);
    run<0, 1>("b32 consecutive bytes (same dword x4)", d); run<1, 1>("b32 stride 4", d); run<3, 1>("b32 random", d);
    return 0;
}
