// orbx_block.h — wave64 / 256-thread workgroup primitives used by the kernels.
#pragma once
#include "orbx_platform.h"

namespace orbx {

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, (unsigned)d);
        if (lane >= d) v += o;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
#ifndef ORBX_EMU
// 32-bit integers: the wave64 inclusive scan as seven DPP adds (row_shr 1,2,3 of the input, row_shr 4 / 8 with bank masks, row_bcast 15 /
// 31 with row masks - the classic GCN sequence) instead of six ds_bpermute round trips through the LDS pipe; lanes that a step does not
// feed add 0.  The sum is the last lane of the scan, read back as a wave-uniform scalar.
__device__ __forceinline__ int wave_incl_scan_i32_dpp(int v) {
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);       // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);       // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, v, 0x113, 0xf, 0xf, false);       // row_shr:3
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xe, false);       // row_shr:4, banks 1-3
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xc, false);       // row_shr:8, banks 2-3
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);       // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);       // row_bcast:31 -> rows 2 and 3
    return x;
}
// 64-bit integers (the workgroup scans of k_quadtree carry two or three 20-bit fields in one value): both halves move through the same DPP
// steps and are added with carry - 28 instructions where six ds_bpermute round trips of both halves took 48 and the latency of the LDS crossbar.
// rows_only: the first five steps, i.e. an inclusive scan inside every row of 16 lanes (all a scan over <= 16 per-wave totals needs).
template <bool rows_only>
__device__ __forceinline__ unsigned long long wave_incl_scan_u64_dpp(unsigned long long v) {
    unsigned long long x = v;
#define ORBX_DPP64(src, ctrl, rmask, bmask) ((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(src), ctrl, rmask, bmask, false) | \
                                             ((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)((src) >> 32), ctrl, rmask, bmask, false) << 32))
    x += ORBX_DPP64(v, 0x111, 0xf, 0xf);
    x += ORBX_DPP64(v, 0x112, 0xf, 0xf);
    x += ORBX_DPP64(v, 0x113, 0xf, 0xf);
    x += ORBX_DPP64(x, 0x114, 0xf, 0xe);
    x += ORBX_DPP64(x, 0x118, 0xf, 0xc);
    if (!rows_only) {
        x += ORBX_DPP64(x, 0x142, 0xa, 0xf);
        x += ORBX_DPP64(x, 0x143, 0xc, 0xf);
    }
#undef ORBX_DPP64
    return x;
}
template <> __device__ __forceinline__ unsigned long long wave_incl_scan<unsigned long long>(unsigned long long v) { return wave_incl_scan_u64_dpp<false>(v); }
template <> __device__ __forceinline__ int wave_incl_scan<int>(int v) { return wave_incl_scan_i32_dpp(v); }
template <> __device__ __forceinline__ unsigned wave_incl_scan<unsigned>(unsigned v) { return (unsigned)wave_incl_scan_i32_dpp((int)v); }
template <> __device__ __forceinline__ int wave_sum<int>(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_i32_dpp(v), 63); }
template <> __device__ __forceinline__ unsigned wave_sum<unsigned>(unsigned v) { return (unsigned)__builtin_amdgcn_readlane(wave_incl_scan_i32_dpp((int)v), 63); }
#endif
// bitwise OR over the wave, the same in every lane (a wave-uniform scalar on the GPU: the DPP sequence of the scan with | for +)
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
#ifdef ORBX_EMU
    for (int d = 32; d >= 1; d >>= 1) v |= __shfl_xor(v, d);
    return v;
#else
    int x = (int)v;
    x |= __builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    x |= __builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    x |= __builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xe, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xc, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (unsigned)__builtin_amdgcn_readlane(x, 63);
#endif
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { unsigned long long o = __shfl_xor(v, d); v = o < v ? o : v; }
    return v;
}
// minimum over the wave, the same in every lane (on the GPU the DPP sequence of the scan with min for +, lanes that a step does not feed
// take the identity: seven v_min_u32_dpp and a lane read where six ds_bpermute round trips went through the LDS crossbar)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#ifdef ORBX_EMU
    for (int d = 32; d >= 1; d >>= 1) { unsigned o = __shfl_xor(v, d); v = o < v ? o : v; }
    return v;
#else
    auto mn = [](unsigned a, unsigned b) { return a < b ? a : b; };
    unsigned x = v;
    x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));
    x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));
    x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x113, 0xf, 0xf, false));
    x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)x, 0x114, 0xf, 0xe, false));
    x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)x, 0x118, 0xf, 0xc, false));
    x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)x, 0x142, 0xa, 0xf, false));
    x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)x, 0x143, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
#endif
}

// Exclusive scan over the workgroup (blockDim.x*blockDim.y*blockDim.z <= 1024, multiple of 64).
// scratch: >= 17 elements of T in LDS.  All threads must call; contains two barriers.
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T* total, T* scratch) {
    const int lane = lane_id();
    const int tid = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
    const int wave = tid >> 6;
    const int nw = (int)((blockDim.x * blockDim.y * blockDim.z + 63) >> 6);
    const T inc = wave_incl_scan(v);
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    T base = 0, tot = 0;
    for (int w = 0; w < nw; w++) { const T s = scratch[w]; if (w < wave) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// The same over the first nw waves of a workgroup whose other waves have exited.
template <typename T>
__device__ __forceinline__ T block_excl_scan_n(T v, T* total, T* scratch, int nw) {
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const T inc = wave_incl_scan(v);
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    T base = 0, tot = 0;
    for (int w = 0; w < nw; w++) { const T s = scratch[w]; if (w < wave) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
#ifndef ORBX_EMU
// 64-bit values on the GPU: the <= 16 per-wave totals are scanned by the first row of lanes of every wave (five DPP steps) and picked out
// with two lane reads, instead of a loop over the waves in every thread (a workgroup of 1024 threads pays ~7 ns per instruction per thread,
// and k_quadtree's critical tree runs a dozen of these scans)
template <>
__device__ __forceinline__ unsigned long long block_excl_scan_n<unsigned long long>(unsigned long long v, unsigned long long* total, unsigned long long* scratch, int nw) {
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned long long inc = wave_incl_scan_u64_dpp<false>(v);
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    const unsigned long long w = wave_incl_scan_u64_dpp<true>(lane < nw ? scratch[lane] : 0ull);
    __syncthreads();
    const unsigned wl = (unsigned)w, wh = (unsigned)(w >> 32);
    const int pw = wave > 0 ? wave - 1 : 0;
    const unsigned long long below = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)wl, pw) | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)wh, pw) << 32);
    *total = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)wl, nw - 1) | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)wh, nw - 1) << 32);
    return (wave > 0 ? below : 0ull) + inc - v;
}
#endif

// a[i] of a small array that lives in registers / kernel arguments, by selects: indexing it with a per-lane value would move the whole
// enclosing struct to private memory (k_frustum and k_lastframe_queries carried a 240-byte scratch segment for F.scale_factors[level])
template <int N> __device__ __forceinline__ float pick(const float (&a)[N], int i) {
    float r = a[0];
#pragma unroll
    for (int k = 1; k < N; k++) r = (i == k) ? a[k] : r;
    return r;
}

}  // namespace orbx
