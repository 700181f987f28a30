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
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { unsigned long long o = __shfl_xor(v, d); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { unsigned o = __shfl_xor(v, d); v = o < v ? o : v; }
    return v;
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

}  // namespace orbx
