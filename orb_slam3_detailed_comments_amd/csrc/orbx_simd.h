// orbx_simd.h — small CDNA byte / packed-16-bit helpers shared by the image kernels (k_image.hip, k_fast.hip), with plain-C equivalents
// for the test emulator (tests/emu).
#pragma once
#include "orbx_types.h"

namespace orbx {

__device__ __forceinline__ int mul24(int a, int b) {
#ifdef ORBX_EMU
    return a * b;
#else
    return __mul24(a, b);      // operands < 2^23: full-rate 24-bit multiply instead of the quarter-rate 32-bit one
#endif
}
// the same, but immune to the compiler turning it back into v_mul_lo_u32 (it does where it has proven narrower operand ranges)
__device__ __forceinline__ int mul24_forced(int a, int b) {
#ifdef ORBX_EMU
    return a * b;
#else
    int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#endif
}
// v_mul_hi_u32_u24: bits 32..47 of the product of the low 24 bits of both operands.  (x * y) >> 16 for x < 2^a, y < 2^b is mulhi_u24(x << (24 - a), y << (24 - b))
// when a + b <= 32: one instruction for the "multiply, keep the high part" of cv::resize's vertical pass
__device__ __forceinline__ uint32_t mulhi_u24(uint32_t a, uint32_t b) {
#ifdef ORBX_EMU
    return (uint32_t)(((unsigned long long)(a & 0xFFFFFFu) * (unsigned long long)(b & 0xFFFFFFu)) >> 32);
#else
    uint32_t r; asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#endif
}
// the same with a wave-uniform first factor (a scalar register as src0 of the VOP2 form: no copy into a vector register)
__device__ __forceinline__ uint32_t mulhi_u24_uniform(uint32_t uniform_a, uint32_t b) {
#ifdef ORBX_EMU
    return (uint32_t)(((unsigned long long)(uniform_a & 0xFFFFFFu) * (unsigned long long)(b & 0xFFFFFFu)) >> 32);
#else
    uint32_t r; asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "s"(uniform_a), "v"(b)); return r;
#endif
}
// v_add3_u32
__device__ __forceinline__ uint32_t add3_u32(uint32_t a, uint32_t b, uint32_t c) {
#ifdef ORBX_EMU
    return a + b + c;
#else
    uint32_t r; asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#endif
}
// ---------------------------------------------------------------------------------------------------
// Small CDNA byte / packed-16-bit helpers (with plain-C equivalents for the test emulator).
// byte permute (v_perm_b32): result byte i = byte sel_i (0..7) of the 8-byte pair {hi:lo}; selector 0x0c gives 0x00
__device__ __forceinline__ uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
#ifdef ORBX_EMU
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t sb = (sel >> (8 * i)) & 0xFF;
        const uint32_t byte = sb >= 0x0c ? 0u : (uint32_t)((v >> (8 * (sb & 7))) & 0xFF);
        r |= byte << (8 * i);
    }
    return r;
#else
    return __builtin_amdgcn_perm(hi, lo, sel);
#endif
}
// v_alignbyte_b32: the 4 bytes starting at byte `shift` (0..3) of the 8-byte pair {hi:lo}
__device__ __forceinline__ uint32_t align_byte(uint32_t hi, uint32_t lo, uint32_t shift) {
#ifdef ORBX_EMU
    return (uint32_t)(((((unsigned long long)hi) << 32) | lo) >> (8 * (shift & 3)));
#else
    return __builtin_amdgcn_alignbyte(hi, lo, shift);
#endif
}
// integer dot products: v_dot4_u32_u8 (four u8 x u8 products + c) and v_dot2_u32_u16 (two u16 x u16 products + c), exact (no clamp)
__device__ __forceinline__ uint32_t dot4_u8(uint32_t a, uint32_t b, uint32_t c) {
#ifdef ORBX_EMU
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
    return c;
#else
    return __builtin_amdgcn_udot4(a, b, c, false);
#endif
}
__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b, uint32_t c) {
#ifdef ORBX_EMU
    return c + (a & 0xFFFFu) * (b & 0xFFFFu) + (a >> 16) * (b >> 16);
#else
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
#endif
}
// v_sad_u8: sum of the four absolute byte differences + c
__device__ __forceinline__ uint32_t sad4_u8(uint32_t a, uint32_t b, uint32_t c) {
#ifdef ORBX_EMU
    for (int i = 0; i < 4; i++) { const int d = (int)((a >> (8 * i)) & 0xFFu) - (int)((b >> (8 * i)) & 0xFFu); c += (uint32_t)(d < 0 ? -d : d); }
    return c;
#else
    return __builtin_amdgcn_sad_u8(a, b, c);
#endif
}
// 4 bytes from any byte address (global and LDS reads need no alignment on gfx950)
__device__ __forceinline__ uint32_t load_u32_any(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
// two signed 16-bit lanes in one VGPR (v_pk_sub_i16 / v_pk_min_i16 / v_pk_max_i16)
#ifdef ORBX_EMU
struct pk2 { short x, y; };
__device__ __forceinline__ pk2 pk_make(uint32_t v) { pk2 r; r.x = (short)(v & 0xFFFF); r.y = (short)(v >> 16); return r; }
__device__ __forceinline__ pk2 pk_sub(pk2 a, pk2 b) { pk2 r; r.x = (short)(a.x - b.x); r.y = (short)(a.y - b.y); return r; }
__device__ __forceinline__ pk2 pk_mad(pk2 a, pk2 b, pk2 c) { pk2 r; r.x = (short)(a.x * b.x + c.x); r.y = (short)(a.y * b.y + c.y); return r; }
__device__ __forceinline__ pk2 pk_min(pk2 a, pk2 b) { pk2 r; r.x = a.x < b.x ? a.x : b.x; r.y = a.y < b.y ? a.y : b.y; return r; }
__device__ __forceinline__ pk2 pk_max(pk2 a, pk2 b) { pk2 r; r.x = a.x > b.x ? a.x : b.x; r.y = a.y > b.y ? a.y : b.y; return r; }
__device__ __forceinline__ int pk_lo(pk2 a) { return a.x; }
__device__ __forceinline__ int pk_hi(pk2 a) { return a.y; }
__device__ __forceinline__ pk2 pk_min3(pk2 a, pk2 b, pk2 c) { return pk_min(pk_min(a, b), c); }
__device__ __forceinline__ pk2 pk_max3(pk2 a, pk2 b, pk2 c) { return pk_max(pk_max(a, b), c); }
__device__ __forceinline__ pk2 pk_bytes(const uint8_t* a, const uint8_t* b) { pk2 r; r.x = (short)*a; r.y = (short)*b; return r; }
__device__ __forceinline__ pk2 pk_xor_or(pk2 a, uint32_t x, uint32_t o) { return pk_make((((uint32_t)(uint16_t)a.x | ((uint32_t)(uint16_t)a.y << 16)) ^ x) | o); }
__device__ __forceinline__ pk2 pk_xor(pk2 a, uint32_t x) { return pk_make(((uint32_t)(uint16_t)a.x | ((uint32_t)(uint16_t)a.y << 16)) ^ x); }
__device__ __forceinline__ uint32_t pk_bits(pk2 a) { return (uint32_t)(uint16_t)a.x | ((uint32_t)(uint16_t)a.y << 16); }
#else
typedef short pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 pk_make(uint32_t v) { return __builtin_bit_cast(pk2, v); }
__device__ __forceinline__ pk2 pk_sub(pk2 a, pk2 b) { return a - b; }
__device__ __forceinline__ pk2 pk_mad(pk2 a, pk2 b, pk2 c) { return a * b + c; }      // v_pk_mad_i16
__device__ __forceinline__ pk2 pk_min(pk2 a, pk2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ pk2 pk_max(pk2 a, pk2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ int pk_lo(pk2 a) { return (int)a.x; }
__device__ __forceinline__ int pk_hi(pk2 a) { return (int)a.y; }
// Three-input packed min / max.  gfx950 has no 3-input packed INTEGER min/max, but it has v_pk_minimum3_f16 / v_pk_maximum3_f16, and positive
// normal binary16 numbers are ordered exactly like their bit patterns read as integers.  Every caller keeps its operands in
// [0x0400, 0x7BFF] (pixel values biased by 0x6400), where the two orders coincide and neither NaN
// nor denormal handling can interfere.  Same issue rate as the 2-input packed ops (tools/valu_issue_microbench.hip).
__device__ __forceinline__ pk2 pk_min3(pk2 a, pk2 b, pk2 c) { pk2 d; asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ pk2 pk_max3(pk2 a, pk2 b, pk2 c) { pk2 d; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// two bytes from two LDS addresses as the two halves of one register; (a ^ x) | o is one v_bitop3_b32
__device__ __forceinline__ pk2 pk_bytes(const uint8_t* a, const uint8_t* b) { return __builtin_bit_cast(pk2, (uint32_t)*a | ((uint32_t)*b << 16)); }   // v_lshl_or_b32 (full rate; v_perm_b32 is not)
__device__ __forceinline__ pk2 pk_xor_or(pk2 a, uint32_t x, uint32_t o) { return __builtin_bit_cast(pk2, (__builtin_bit_cast(uint32_t, a) ^ x) | o); }
__device__ __forceinline__ pk2 pk_xor(pk2 a, uint32_t x) { return __builtin_bit_cast(pk2, __builtin_bit_cast(uint32_t, a) ^ x); }
__device__ __forceinline__ uint32_t pk_bits(pk2 a) { return __builtin_bit_cast(uint32_t, a); }
#endif
struct u32x2 { uint32_t lo, hi; };
// 8 bytes from any byte address of the LDS / global memory (one ds_read_b64 / global_load_dwordx2; no alignment requirement on gfx950)
__device__ __forceinline__ u32x2 load_u64_any(const uint8_t* p) { u32x2 v; __builtin_memcpy(&v, p, 8); return v; }
// Buffer addressing (buffer_load_dword v, voffset, s[rsrc], soffset offen): a wave-uniform base (resource descriptor in SGPRs) plus a
// wave-uniform byte offset (SGPR) plus a per-lane byte offset (VGPR) - streaming kernels that walk rows pay no vector instruction per
// address.  Raw (stride 0), offsets unchecked up to 2 GiB.
#ifdef ORBX_EMU
struct BufRsrc { uint8_t* p; };
__device__ __forceinline__ BufRsrc buf_make(const void* p) { BufRsrc r; r.p = (uint8_t*)p; return r; }
__device__ __forceinline__ uint32_t buf_load_u32(BufRsrc r, uint32_t voff, uint32_t soff) { uint32_t v; __builtin_memcpy(&v, r.p + voff + soff, 4); return v; }
__device__ __forceinline__ void buf_store_u32(uint32_t v, BufRsrc r, uint32_t voff, uint32_t soff) { __builtin_memcpy(r.p + voff + soff, &v, 4); }
__device__ __forceinline__ u32x2 buf_load_u64(BufRsrc r, uint32_t voff, uint32_t soff) { u32x2 v; __builtin_memcpy(&v, r.p + voff + soff, 8); return v; }
#else
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc buf_make(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7FFFFFFF, 0x00020000); }
__device__ __forceinline__ uint32_t buf_load_u32(BufRsrc r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0); }
__device__ __forceinline__ void buf_store_u32(uint32_t v, BufRsrc r, uint32_t voff, uint32_t soff) { __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)voff, (int)soff, 0); }
// 8 bytes from any byte address (no alignment requirement on gfx950)
__device__ __forceinline__ u32x2 buf_load_u64(BufRsrc r, uint32_t voff, uint32_t soff) {
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0); u32x2 o; o.lo = v.x; o.hi = v.y; return o;
}
#endif
// ---------------------------------------------------------------------------------------------------
// v_mfma_i32_32x32x32_i8: D (32 x 32, i32) += A (32 x 32, i8) * B (32 x 32, i8).  Lane l holds 16 consecutive k of row l & 31 of A (k block
// l >> 5) and 16 consecutive k of column l & 31 of B; D: column l & 31, rows (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) for register r < 16.
#ifdef ORBX_EMU
struct v4i_t { int e[4]; };
struct v16i_t { int e[16]; int& operator[](int i) { return e[i]; } int operator[](int i) const { return e[i]; } };
__device__ __forceinline__ v16i_t v16i_zero() { v16i_t z; for (int i = 0; i < 16; i++) z.e[i] = 0; return z; }
__device__ __forceinline__ v4i_t load_v4i(const uint8_t* p) { v4i_t v; __builtin_memcpy(&v, p, 16); return v; }
inline v16i_t mfma_i8_32x32x32(v4i_t a, v4i_t b, v16i_t c) {
    uint64_t m[2][2], A[2][hipemu::kWave], Bv[2][hipemu::kWave];
    __builtin_memcpy(m[0], &a, 16); __builtin_memcpy(m[1], &b, 16);
    hipemu::wave_exchange(m[0][0], A[0]); hipemu::wave_exchange(m[0][1], A[1]);
    hipemu::wave_exchange(m[1][0], Bv[0]); hipemu::wave_exchange(m[1][1], Bv[1]);
    const int lane = hipemu::cur().lane, col = lane & 31;
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int sum = 0;
        for (int h = 0; h < 2; h++)
            for (int j = 0; j < 16; j++) {
                const int x = (int)(signed char)((A[j >> 3][row + 32 * h] >> (8 * (j & 7))) & 0xFF);
                const int y = (int)(signed char)((Bv[j >> 3][col + 32 * h] >> (8 * (j & 7))) & 0xFF);
                sum += x * y;
            }
        c.e[r] += sum;
    }
    return c;
}
#else
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ v16i_t v16i_zero() { v16i_t z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; return z; }
__device__ __forceinline__ v4i_t load_v4i(const uint8_t* p) { return *(const v4i_t*)p; }
__device__ __forceinline__ v16i_t mfma_i8_32x32x32(v4i_t a, v4i_t b, v16i_t c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
#endif
__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned umax32(unsigned a, unsigned b) { return a > b ? a : b; }
constexpr int kPixBias = 0x6400;     // pixel value b is carried as 0x6400 + b (binary16 1024 + b)

}  // namespace orbx
