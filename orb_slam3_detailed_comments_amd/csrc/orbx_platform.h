// orbx_platform.h — the one place that knows whether the kernel sources are being compiled by hipcc
// for gfx950 (the product) or by g++ against tests/emu/hip_emu.h (CPU SIMT emulator, tests only).
#pragma once
#include <cstddef>
#include <cstdint>

#ifdef ORBX_EMU
#include "hip_emu.h"   // tests/emu, via -I
#define ORBX_DYN_SMEM(name) unsigned char* name = hipemu::cur().blk->dyn_smem
#define ORBX_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipemu::launch((grid), (block), (smem), [=]() { kern(__VA_ARGS__); })
#define ORBX_HD
// lanes of a wave run in lockstep on the GPU; the emulator's fibers do not, so code that relies on "every lane has read
// before any lane writes" marks the point explicitly
#define ORBX_WAVE_SYNC() hipemu::wave_barrier()
#define ORBX_READLANE(v, l) __shfl((v), (l))
#define ORBX_UNIFORM(v) (v)
#define ORBX_BALLOT(pred) __ballot((pred) ? 1 : 0)
#define ORBX_IN_BALLOT(mask) ((((mask) >> (threadIdx.x & 63u)) & 1ull) != 0ull)
#else
#include <hip/hip_runtime.h>
#define ORBX_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define ORBX_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__)
#define ORBX_HD __host__ __device__
#define ORBX_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// value of lane l as a wave-uniform scalar (SGPR): keeps counters derived from it out of the vector registers
#define ORBX_READLANE(v, l) __builtin_amdgcn_readlane((v), (l))
// a value the code knows to be the same in every lane of the wave (e.g. derived from threadIdx.y of a (64, n) block), moved to an SGPR so that
// what is computed from it runs on the scalar unit
#define ORBX_UNIFORM(v) __builtin_amdgcn_readfirstlane(v)
// wave64 ballot of a PREDICATE: llvm.amdgcn.ballot.i64 takes the i1 itself, so the comparison that produces the predicate writes the SGPR mask
// directly.  HIP's __ballot(int) compares a materialised integer with zero: the predicate is first turned into 0 / 1 (v_bfe / v_cndmask) and then
// compared again - two VOP3 instructions per ballot on top of the test itself (seen in k_fast_cells' append code: 16 of them per trip).
#define ORBX_BALLOT(pred) __builtin_amdgcn_ballot_w64((bool)(pred))
// "is my lane in this ballot": the SGPR mask itself becomes the exec mask of the guarded code (llvm.amdgcn.inverse.ballot) - one test serves
// the ballot, its popcount, the mbcnt ranks and the branch
#define ORBX_IN_BALLOT(mask) __builtin_amdgcn_inverse_ballot_w64(mask)
#endif

namespace orbx {
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }
// lane l's 64-bit value as a wave-uniform scalar pair (base addresses then live in SGPRs and loads use 32-bit vector offsets)
__device__ __forceinline__ long long readlane_i64(long long v, int l) {
#ifdef ORBX_EMU
    return __shfl(v, l);
#else
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFFll), l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return ((long long)hi << 32) | (unsigned long long)(unsigned)lo;
#endif
}
}  // namespace orbx
