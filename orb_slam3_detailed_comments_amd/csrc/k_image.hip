// k_image.hip — the image-space (streaming) kernels of the ORB extractor for gfx950:
//   k_resize      pyramid level l from level l-1   (reference ComputePyramid, src/ORBextractor.cc:1687-1738,
//                                                    arithmetic of cv::resize INTER_LINEAR 8U)
//   k_fast_cells  per-cell FAST-9/16 score + cell-local 3x3 NMS + per-cell threshold choice + ordered
//                 compaction               (reference ComputeKeyPointsOctTree, :1061-1166, cv::FAST)
//   k_blur        7x7 sigma=2 Gaussian, REFLECT_101, 8-bit fixed point        (:1629-1637, cv::GaussianBlur)
// All are HBM/LDS-bound integer kernels: no MFMA.  One launch covers every image of the batch.
#include "orbx_types.h"
#include "orbx_block.h"
#include "orbx_kernels.h"

namespace orbx {

// ---------------------------------------------------------------------------------------------------
// Level 0 import: user images (arbitrary stride, device-visible) -> pyramid level 0 (64-byte-multiple pitch).
// block (64,4): 256 bytes x 4 rows.  grid (ceil(pitch0/256), ceil(h/4), B)
__global__ void __launch_bounds__(256) k_import(const LevelInfo* __restrict__ lv, const uint8_t* __restrict__ images,
                                                int stride, size_t image_stride, uint8_t* __restrict__ pyr, size_t pyr_stride) {
    const LevelInfo D = lv[0];
    const int b = (int)blockIdx.z;
    const int y = (int)(blockIdx.y * 4 + threadIdx.y);
    const int x0 = (int)(blockIdx.x * 64 + threadIdx.x) * 4;
    if (y >= D.h || x0 >= D.pitch) return;
    const uint8_t* src = images + (size_t)b * image_stride + (size_t)y * stride;
    uint32_t out = 0;
    if ((((size_t)(src + x0)) & 3) == 0 && x0 + 3 < D.w) out = *(const uint32_t*)(src + x0);   // aligned interior dword
    else {
#pragma unroll
        for (int k = 0; k < 4; k++) if (x0 + k < D.w) out |= (uint32_t)src[x0 + k] << (8 * k);
    }
    *(uint32_t*)(pyr + (size_t)b * pyr_stride + D.off + (size_t)y * D.pitch + x0) = out;
}

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
// ---------------------------------------------------------------------------------------------------
// Pyramid: cv::resize INTER_LINEAR 8U, 11-bit fixed point.  A workgroup produces a 256 x 8 output tile: the source
// rows/columns it needs (<= ~1.2x the tile for the 1.2 pyramid) are staged in LDS with aligned dword loads, each
// thread then interpolates 4 adjacent outputs on 2 rows and stores them as dwords.
// block (64,4); grid (ceil(pitch/256), ceil(h/8), B); dynamic LDS = lds_rows * lds_pitch bytes.
__global__ void __launch_bounds__(256) k_resize(const LevelInfo* __restrict__ lv, int level,
                                                const ResizeTap* __restrict__ xtab,
                                                const ResizeTap* __restrict__ ytab,
                                                uint8_t* __restrict__ pyr, size_t pyr_stride, int lds_pitch, int lds_rows) {
    ORBX_DYN_SMEM(smem);
    const LevelInfo D = lv[level];
    const LevelInfo S = lv[level - 1];
    const int b = (int)blockIdx.z;
    const int tid = (int)(threadIdx.y * 64 + threadIdx.x);
    const int dxb = (int)blockIdx.x * 256, dyb = (int)blockIdx.y * kResizeRows;
    const uint8_t* src = pyr + (size_t)b * pyr_stride + S.off;
    uint8_t* dst = pyr + (size_t)b * pyr_stride + D.off;
    const ResizeTap* xt = xtab + D.xtab_off;
    const ResizeTap* yt = ytab + D.ytab_off;
    // source window of this tile (block-uniform)
    const int dx_last = imin(dxb + 255, D.w - 1), dy_last = imin(dyb + kResizeRows - 1, D.h - 1);
    const int gx0 = xt[imin(dxb, D.w - 1)].ofs & ~3;
    const int sx_hi = imin(xt[dx_last].ofs + 1, S.w - 1);
    const int ncd = imin(((sx_hi - gx0) >> 2) + 1, lds_pitch >> 2);
    const int sy_lo = imin(imax(yt[imin(dyb, D.h - 1)].ofs, 0), S.h - 1);
    const int sy_hi = imin(imax(yt[dy_last].ofs + 1, 0), S.h - 1);
    const int nrow = imin(sy_hi - sy_lo + 1, lds_rows);
    // this thread's interpolation taps (4 columns, 2 rows): loaded up front so that their latency overlaps the window loads
    const int dx0 = dxb + (int)threadIdx.x * 4;
    ResizeTap txs[4], tys[kResizeRows / 4];
#pragma unroll
    for (int k = 0; k < 4; k++) txs[k] = xt[imin(dx0 + k, D.w - 1)];
#pragma unroll
    for (int rr = 0; rr < kResizeRows / 4; rr++) tys[rr] = yt[imin(dyb + (int)threadIdx.y + 4 * rr, D.h - 1)];
    {   // the window is a few dwords per thread: all of a thread's global loads are issued before the first LDS store
        const int n = nrow * ncd;
        const unsigned Mc = (1u << 20) / (unsigned)ncd + 1u;     // i / ncd == (i * Mc) >> 20 exactly for i < 2^13
        for (int i0 = tid; i0 < n; i0 += 1024) {
            uint32_t v[4]; int o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = i0 + 256 * k;
                o[k] = -1;
                if (i < n) {
                    const int r = (int)((unsigned)mul24(i, (int)Mc) >> 20), c = i - mul24(r, ncd);
                    o[k] = mul24(r, lds_pitch) + 4 * c;
                    v[k] = *(const uint32_t*)(src + mul24(sy_lo + r, S.pitch) + gx0 + 4 * c);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) if (o[k] >= 0) *(uint32_t*)(smem + o[k]) = v[k];
        }
    }
    __syncthreads();
    if (dx0 >= D.pitch) return;
    int sxo[4], sxo1[4], a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const ResizeTap tx = txs[k];
        sxo[k] = tx.ofs - gx0; sxo1[k] = imin(tx.ofs + 1, S.w - 1) - gx0;
        a0[k] = (int)(int16_t)(tx.w & 0xFFFF); a1[k] = tx.w >> 16;
    }
#pragma unroll
    for (int rr = 0; rr < kResizeRows / 4; rr++) {
        const int dy = dyb + (int)threadIdx.y + 4 * rr;
        if (dy >= D.h) break;
        const ResizeTap ty = tys[rr];
        const int r0 = imin(imax(ty.ofs, 0), S.h - 1) - sy_lo, r1 = imin(imax(ty.ofs + 1, 0), S.h - 1) - sy_lo;
        const int b0 = (int)(int16_t)(ty.w & 0xFFFF), b1 = ty.w >> 16;
        const uint8_t* S0 = smem + mul24(r0, lds_pitch);
        const uint8_t* S1 = smem + mul24(r1, lds_pitch);
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (dx0 + k < D.w) {
                // every factor is below 2^23 and every product below 2^31 (pixel < 2^8, tap <= 2^11, h >> 4 < 2^16): 24-bit multiplies,
                // the 32-bit v_mul_lo_u32 runs at a quarter of their rate
                const int h0 = mul24((int)S0[sxo[k]], a0[k]) + mul24((int)S0[sxo1[k]], a1[k]);
                const int h1 = mul24((int)S1[sxo[k]], a0[k]) + mul24((int)S1[sxo1[k]], a1[k]);
                int v = ((mul24_forced(b0, h0 >> 4) >> 16) + (mul24_forced(b1, h1 >> 4) >> 16) + 2) >> 2;
                v = imin(imax(v, 0), 255);
                out |= (uint32_t)v << (8 * k);
            }
        }
        *(uint32_t*)(dst + mul24(dy, D.pitch) + dx0) = out;
    }
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
#endif
constexpr int kPixBias = 0x6400;     // pixel value b is carried as 0x6400 + b (binary16 1024 + b)

// ---------------------------------------------------------------------------------------------------
// FAST-9/16.  ring offsets (dx,dy), k = 0..15, as in OpenCV: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)
// (0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3).  d[k] = v - ring[k].
// One workgroup per (cell, image).  LDS: window tile (dword-aligned columns) | score tile | candidate list (u16).
//   A  every interior pixel: ring + exact quick rejection; survivors are appended, in row-major order, to the list
//      of the wave that owns that quarter of the pixel range (wave ballots, no barrier)
//   B  full corner score only for listed pixels (all lanes busy)
//   C  cell-local strict 3x3 NMS on listed pixels; does any survivor reach iniTh?
//   D  threshold choice of the reference (FAST at iniTh; if that yields nothing, FAST at minTh, :1135-1148) and
//      ordered compaction into the cell's slot list.
// slots: per-cell candidate lists in the reference order; cell_count[b*ncells + cell] = number kept.
// OpenCV's cornerScore (largest threshold for which the pixel is still a corner) when the pixel is a corner at threshold t0, else 0:
// max over the 16 nine-arcs of min |v - ring|, minus 1 - sliding 9-window minima by doubling, two pixels per instruction (packed 16-bit).
// one-sided variant: d[k] = sign * (v - ring_k) with sign = +1 for a dark candidate and -1 for a bright one, so that both become
// "max over the 16 nine-arcs of the arc minimum"
// x[k] = ring value of a dark candidate, 255 - ring value of a bright one (+ kPixBias), vp = the centre treated the same way:
//   dark   max over 9-arcs of min (v - r)  =  v - min over arcs of max r
//   bright max over 9-arcs of min (r - v)  = (255 - v) - min over arcs of max (255 - r)
// so both polarities are "centre minus the smallest 9-arc maximum": 9-arc maxima as max3 of three 3-arc maxima, the minimum over the 16
// arcs by min3.
__device__ __forceinline__ void fast_score_pk(const pk2 x[16], pk2 vp, int t0, int& sA, int& sB) {
    pk2 w3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w3[k] = pk_max3(x[k], x[(k + 1) & 15], x[(k + 2) & 15]);
    pk2 w9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w9[k] = pk_max3(w3[k], w3[(k + 3) & 15], w3[(k + 6) & 15]);
    pk2 m[5];
#pragma unroll
    for (int k = 0; k < 5; k++) m[k] = pk_min3(w9[3 * k], w9[3 * k + 1], w9[3 * k + 2]);
    const pk2 W = pk_min3(pk_min3(m[0], m[1], m[2]), pk_min3(m[3], m[4], w9[15]), w9[15]);
    const pk2 df = pk_sub(vp, W);
    const int mA = pk_lo(df), mB = pk_hi(df);
    sA = mA > t0 ? mA - 1 : 0;
    sB = mB > t0 ? mB - 1 : 0;
}

constexpr int kFastWaves = 1;                       // waves per FAST workgroup (one cell per workgroup)
constexpr int kFastThreads = 64 * kFastWaves;
// survivor-list entry: score-tile index (13 bits) | bright-candidate flag | "second entry of the same pixel" flag | NMS-keep flag
constexpr int kListPos = 0x1FFF, kListBright = 0x2000, kListDup = 0x4000;

__global__ void __launch_bounds__(kFastThreads) k_fast_cells(const LevelInfo* __restrict__ lv,
                                                    const CellInfo* __restrict__ cells, int ncells,
                                                    const uint8_t* __restrict__ pyr, size_t pyr_stride,
                                                    int iniTh, int minTh,
                                                    uint32_t* __restrict__ slots, size_t slots_stride,
                                                    int* __restrict__ cell_count, int tile_bytes, int inner_bytes, int list_bytes) {
    ORBX_DYN_SMEM(smem);
    // Workgroup -> cell mapping.  Consecutive workgroup ids go to different XCDs (id % 8), each with its own L2; with the plain mapping
    // (workgroup b -> cell b) neighbouring cells never share an L2 and the 6-pixel window overlap plus the dword / cache-line padding of
    // every 43-byte window row is fetched again per cell.  Runs of kFastXcdRun neighbouring cells are therefore kept on one XCD.
    // Measured per 128-image launch, run length 1 / 2 / 4 / 20: 305 / 186 / 131 / 80 MB fetched, 0.458 / 0.485 / 0.475 / 0.560 ms,
    // 63.0 / 62.6 / 62.3 / 58.2 k pairs/s end to end - the kernel is not memory-bound and long runs skew the mix of dense and sparse cells
    // per XCD, so the run stays short: 4 brings the traffic down to the algorithmic bytes for 1 % of throughput.
    const int bx = (int)blockIdx.x, xcd = bx & 7, jj = bx >> 3;
    const int cell = ((jj / kFastXcdRun) * 8 + xcd) * kFastXcdRun + (jj % kFastXcdRun), b = (int)blockIdx.y;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    if (cell >= ncells) return;
    const CellInfo ci = cells[cell];
    const LevelInfo L = lv[ci.level];
    const int iw = ci.x1 - ci.x0, ih = ci.y1 - ci.y0;
    if (iw <= 0 || ih <= 0) {
        if (tid == 0) cell_count[(size_t)b * ncells + cell] = 0;
        return;
    }
    const int wh = ih + 6;
    const int gx0 = ((ci.x0 - 3) & ~3) - 4, gx1 = ((ci.x1 + 3 + 3) & ~3) + 4;   // dword-aligned window + one dword margin each side
    // (a compile-time tile pitch - 64/128 or 68/132 bytes - makes the LDS offsets immediates but measured 3-10 % slower: more LDS per
    //  workgroup and, for 64, row-on-row bank conflicts; the runtime pitch stays)
    const int wpd = (gx1 - gx0) >> 2, wp = wpd * 4;
    const int xo = (ci.x0 - 3) - gx0;
    uint8_t* tile = smem;
    uint8_t* sc = smem + tile_bytes;
    uint16_t* list = (uint16_t*)(sc + inner_bytes);
#ifdef ORBX_FAST_LIST_CAP                                    // tests rebuild with a tiny capacity to force the flush path
    const int list_cap = ORBX_FAST_LIST_CAP;
#else
    const int list_cap = list_bytes >> 1;                    // entries
#endif
    const uint8_t* img = pyr + (size_t)b * pyr_stride + L.off;
    const unsigned Mw = (1u << 20) / (unsigned)wpd + 1u;    // i / wpd == (i * Mw) >> 20 exactly for i < 2^13
    for (int i0 = tid; i0 < wh * wpd; i0 += 4 * kFastThreads) {          // four loads in flight per lane before the first LDS store
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + k * kFastThreads;
            if (i < wh * wpd) {
                const int r = (int)((unsigned)mul24(i, (int)Mw) >> 20), c = i - mul24(r, wpd);
                v[k] = *(const uint32_t*)(img + (uint32_t)(mul24(ci.y0 - 3 + r, L.pitch) + gx0 + 4 * c));      // uniform base + 32-bit offset
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { const int i = i0 + k * kFastThreads; if (i < wh * wpd) ((uint32_t*)tile)[i] = v[k]; }
    }
    for (int i = tid; i < (inner_bytes >> 2); i += kFastThreads) ((uint32_t*)sc)[i] = 0u;   // scores default to 0 (not a corner)
    __syncthreads();
    const int t0 = imin(iniTh, minTh);
    // scores live in a tile with a one-pixel zero frame (pitch iw + 2), so the 3x3 NMS reads its 8 neighbours at fixed offsets
    // without bounds tests; list entries are indices into that tile
    const int pitch = iw + 2;
    const unsigned M = (1u << 20) / (unsigned)pitch + 1u;   // p / pitch == (p * M) >> 20 exactly for p < 2^13, pitch <= 128
    const unsigned long long lt = (1ull << lane) - 1ull;
    // ---- A ----  work item = (row y, dword group g): 4 adjacent pixels per lane, packed 16-bit arithmetic.
    // The 16 ring bytes of the 4 pixels are cut out of 21 aligned LDS dwords with v_alignbyte, widened with v_perm,
    // and the exact opposite-pair rejection runs on two pixels per instruction (v_pk_sub/min/max_i16).
    // The survivor list holds list_cap entries (half of the cell's pixels; LDS per workgroup sets the occupancy of this kernel):
    // when the next trip would overflow it, the listed pixels are scored at once (phase B) and the list restarts.  Cells that
    // needed such a flush (dense noise) finish with the list-free variants of phases C and D.
    static_assert(kFastWaves == 1, "the flush logic below assumes one wave per cell");
    const int g0 = (xo + 3) >> 2, ng = ((xo + 3 + iw - 1) >> 2) - g0 + 1;
    const uint32_t* tile32 = (const uint32_t*)tile;
    // ---- B ----  full score of list[0, n), two listed pixels per lane (packed lanes)
    auto score_listed = [&](int n) {
        for (int i = 2 * lane; i < n; i += 2 * kFastThreads) {
            const int eA = list[i];
            const bool hasB = i + 1 < n;
            const int eB = hasB ? list[i + 1] : eA;
            const int pA = eA & kListPos, pB = eB & kListPos;
            // polarity of the entry: a bright candidate is scored on the inverted image (x ^ 0xFF), see fast_score_pk
            const uint32_t xm = ((eA & kListBright) ? 0x000000FFu : 0u) | ((eB & kListBright) ? 0x00FF0000u : 0u);
            const uint32_t bias2 = (uint32_t)kPixBias * 0x00010001u;
            // (24-bit multiplies: every operand here is far below 2^23 and the products below 2^31; v_mul_lo_u32 runs at quarter rate)
            const int yA = (int)((unsigned)mul24(pA, (int)M) >> 20), xA = pA - mul24(yA, pitch);     // (y + 1, x + 1)
            const int yB = (int)((unsigned)mul24(pB, (int)M) >> 20), xB = pB - mul24(yB, pitch);
            const uint8_t* cA = tile + mul24(yA + 2, wp) + xo + xA + 2;
            const uint8_t* cB = tile + mul24(yB + 2, wp) + xo + xB + 2;
            // one pointer per window row and pixel, so that the 16 ring reads are immediate offsets from them
            // (gathering the ring with 7 unaligned ds_read_b32/b64 per pixel instead of 17 byte reads was measured 67 % slower; packing the two
            //  pixels' bytes with ds_read_u8_d16 / _d16_hi does not work on this part: with SRAM ECC enabled the d16 loads clear the other half)
            const uint8_t *a0 = cA - 3 * wp, *a1 = cA - 2 * wp, *a2 = cA - wp, *a4 = cA + wp, *a5 = cA + 2 * wp, *a6 = cA + 3 * wp;
            const uint8_t *b0 = cB - 3 * wp, *b1 = cB - 2 * wp, *b2 = cB - wp, *b4 = cB + wp, *b5 = cB + 2 * wp, *b6 = cB + 3 * wp;
            const pk2 vp = pk_xor_or(pk_bytes(cA, cB), xm, bias2);
            pk2 d[16];
#define ORBX_D(k, ra, rb, dx) d[k] = pk_xor_or(pk_bytes(ra + (dx), rb + (dx)), xm, bias2);
            ORBX_D(0, a6, b6, 0)    ORBX_D(1, a6, b6, 1)    ORBX_D(2, a5, b5, 2)    ORBX_D(3, a4, b4, 3)
            ORBX_D(4, cA, cB, 3)    ORBX_D(5, a2, b2, 3)    ORBX_D(6, a1, b1, 2)    ORBX_D(7, a0, b0, 1)
            ORBX_D(8, a0, b0, 0)    ORBX_D(9, a0, b0, -1)   ORBX_D(10, a1, b1, -2)  ORBX_D(11, a2, b2, -3)
            ORBX_D(12, cA, cB, -3)  ORBX_D(13, a4, b4, -3)  ORBX_D(14, a5, b5, -2)  ORBX_D(15, a6, b6, -1)
#undef ORBX_D
            int sA, sB;
            fast_score_pk(d, vp, t0, sA, sB);
            // a pixel that passed the quick test for both polarities has two entries; it can be a corner for at most one of them
            // (two 9-arcs of opposite sign do not fit on 16 ring pixels), so only a positive score is written
            if (sA > 0) sc[pA] = (uint8_t)sA;
            if (hasB && sB > 0) sc[pB] = (uint8_t)sB;
        }
        ORBX_WAVE_SYNC();
    };
    int cnt = 0;
    bool flushed = false;
    {
        // A trip covers rpt = 64 / ng whole rows: lane = (row within the trip) * ng + (dword group), so a lane keeps its group - and with it
        // its x position, the validity of its four pixels and its tile column - for the whole cell, and only the row advances.  Row-major
        // order of the lanes is row-major order of the pixels.  (ng <= 11 for the cell sizes in use: at most 14 % of the lanes idle.)
        const int rpt = imax(64 / ng, 1);                          // (a row of more than 64 groups = 256 pixels does not occur: cells are < 70 px wide)
        const int yl = lane / ng, gi = lane - yl * ng;
        const bool lane_used = yl < rpt;
        const int g = g0 + gi;
        const int xbase = 4 * g - (xo + 3);                    // interior x of this lane's first pixel (may be < 0)
        const int vlo_ = imax(0, -xbase), vhi_ = imin(4, iw - xbase);          // valid pixels j in [vlo_, vhi_)
        const bool v0 = lane_used && vlo_ <= 0 && vhi_ > 0, v1 = lane_used && vlo_ <= 1 && vhi_ > 1, v2 = lane_used && vlo_ <= 2 && vhi_ > 2, v3 = lane_used && vhi_ > 3 && vlo_ <= 3;
        const uint32_t* rp_lane = tile32 + mul24(yl + 3, wpd) + g;
        const int pj_lane = mul24(yl + 1, pitch) + xbase + 1;
        // outer loop: one turn per list fill.  The trip that would overflow the list is abandoned (nothing appended, the (row, group)
        // cursor not advanced), the listed pixels are scored, and the same trip is redone with an empty list - one call site of
        // phase B outside the phase-A loop keeps the register budgets of the two phases apart.
        int it0 = 0;                                               // first row of the current trip
        for (;;) {
        for (; it0 < ih; it0 += rpt) {
            const int y = it0 + yl;
            // survivor predicates of this lane's 4 pixels (dark / bright): kept as eight lane masks, so that the appends below run under them
            // directly instead of re-testing bits of a per-lane bit mask
            bool pd0 = false, pd1 = false, pd2 = false, pd3 = false, pb0 = false, pb1 = false, pb2 = false, pb3 = false;
            if (lane_used && y < ih) {
                const uint32_t* rp = rp_lane + mul24(it0, wpd);
                const uint32_t kBias4 = (uint32_t)(kPixBias >> 8) * 0x01010101u;          // the high byte of every widened pixel
                // Only the four opposite pairs (0,8) (2,10) (4,12) (6,14) are tested here: still a necessary condition (every 9-arc holds one
                // point of each opposite pair), it lets 34 % instead of 31 % of the pixels through to the exact score of phase B, and it costs
                // half the ring extraction and pair arithmetic of the full eight-pair test (8 ring points from 11 LDS dwords instead of 16 from 21).
                // Measured per 128 images: eight pairs 0.415 ms, four 0.396 ms, the two compass pairs alone 0.409 ms (41 % survivors).
                const uint32_t* r0 = rp - 3 * wpd; const uint32_t* r1 = rp - 2 * wpd; const uint32_t* r5 = rp + 2 * wpd; const uint32_t* r6 = rp + 3 * wpd;
                const uint32_t C0 = r0[0], C6 = r6[0], L1 = r1[-1], C1 = r1[0], R1 = r1[1], L3 = rp[-1], C3 = rp[0], R3 = rp[1], L5 = r5[-1], C5 = r5[0], R5 = r5[1];
                pk2 rlo[8], rhi[8];          // index = k / 2 for k = 0, 2, .., 14
#define ORBX_RING(i, w4) { const uint32_t w = (w4); rlo[i] = pk_make(byte_perm(kBias4, w, 0x04010400u)); rhi[i] = pk_make(byte_perm(kBias4, w, 0x04030402u)); }
                ORBX_RING(0, C6)                          // k = 0   ( 0, +3)
                ORBX_RING(1, align_byte(R5, C5, 2))       // k = 2   (+2, +2)
                ORBX_RING(2, align_byte(R3, C3, 3))       // k = 4   (+3,  0)
                ORBX_RING(3, align_byte(R1, C1, 2))       // k = 6   (+2, -2)
                ORBX_RING(4, C0)                          // k = 8   ( 0, -3)
                ORBX_RING(5, align_byte(C1, L1, 2))       // k = 10  (-2, -2)
                ORBX_RING(6, align_byte(C3, L3, 1))       // k = 12  (-3,  0)
                ORBX_RING(7, align_byte(C5, L5, 2))       // k = 14  (-2, +2)
#undef ORBX_RING
                // a dark 9-arc needs min(ring_k, ring_k+8) < v - t for every opposite pair, a bright one max(..) > v + t:
                //   M = max_k min(pair) ,  N = min_k max(pair) ;  possible corner  =>  v - M > t  or  N - v > t
                pk2 mn_lo[4], mx_lo[4], mn_hi[4], mx_hi[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    mn_lo[k] = pk_min(rlo[k], rlo[k + 4]); mx_lo[k] = pk_max(rlo[k], rlo[k + 4]);
                    mn_hi[k] = pk_min(rhi[k], rhi[k + 4]); mx_hi[k] = pk_max(rhi[k], rhi[k + 4]);
                }
                const pk2 M_lo = pk_max(pk_max3(mn_lo[0], mn_lo[1], mn_lo[2]), mn_lo[3]), M_hi = pk_max(pk_max3(mn_hi[0], mn_hi[1], mn_hi[2]), mn_hi[3]);
                const pk2 N_lo = pk_min(pk_min3(mx_lo[0], mx_lo[1], mx_lo[2]), mx_lo[3]), N_hi = pk_min(pk_min3(mx_hi[0], mx_hi[1], mx_hi[2]), mx_hi[3]);
                const pk2 vlo = pk_make(byte_perm(kBias4, C3, 0x04010400u)), vhi = pk_make(byte_perm(kBias4, C3, 0x04030402u));
                const pk2 dk_lo = pk_sub(vlo, M_lo), dk_hi = pk_sub(vhi, M_hi), br_lo = pk_sub(N_lo, vlo), br_hi = pk_sub(N_hi, vhi);
                pd0 = v0 && pk_lo(dk_lo) > t0; pd1 = v1 && pk_hi(dk_lo) > t0; pd2 = v2 && pk_lo(dk_hi) > t0; pd3 = v3 && pk_hi(dk_hi) > t0;
                pb0 = v0 && pk_lo(br_lo) > t0; pb1 = v1 && pk_hi(br_lo) > t0; pb2 = v2 && pk_lo(br_hi) > t0; pb3 = v3 && pk_hi(br_hi) > t0;
            }
            const int c4 = (int)pd0 + (int)pd1 + (int)pd2 + (int)pd3 + (int)pb0 + (int)pb1 + (int)pb2 + (int)pb3;
            const int incl = wave_incl_scan(c4);
            const int trip = ORBX_READLANE(incl, 63);
            if (cnt > 0 && cnt + trip > list_cap) break;                                    // wave-uniform (a trip adds <= 512 entries <= list_cap)
            int pos = cnt + incl - c4;
            const int pj0 = pj_lane + mul24(it0, pitch);
#define ORBX_APPEND(j, pd, pb) \
            if (pd) list[pos++] = (uint16_t)(pj0 + (j)); \
            if (pb) list[pos++] = (uint16_t)((pj0 + (j)) | kListBright | ((pd) ? kListDup : 0));
            ORBX_APPEND(0, pd0, pb0) ORBX_APPEND(1, pd1, pb1) ORBX_APPEND(2, pd2, pb2) ORBX_APPEND(3, pd3, pb3)
#undef ORBX_APPEND
            cnt += trip;
        }
        ORBX_WAVE_SYNC();
        score_listed(cnt);
        if (it0 >= ih) break;
        cnt = 0; flushed = true;
        }
    }
    const int total = cnt;
    uint32_t* out = slots + (size_t)b * slots_stride + ci.slot_off;
    int base = 0;
    if (!flushed) {
        // ---- C ----  cell-local strict 3x3 NMS on the listed pixels
        int any_hi = 0;
        for (int i = lane; i < total; i += kFastThreads) {
            const int e = list[i];
            const int p = e & kListPos;
            const uint8_t* c = sc + p;
            const int s = (e & kListDup) ? 0 : c[0];          // the second entry of a two-polarity pixel is not a second pixel
            // neighbours outside the cell interior are the zero frame; s == 0 (not a corner) fails every strict comparison
            const int m0 = imax(imax((int)c[-pitch - 1], (int)c[-pitch]), imax((int)c[-pitch + 1], (int)c[-1]));
            const int m1 = imax(imax((int)c[1], (int)c[pitch - 1]), imax((int)c[pitch], (int)c[pitch + 1]));
            const int keep = s > imax(m0, m1);
            if (keep) list[i] = (uint16_t)(e | 0x8000);
            any_hi |= (keep && s >= iniTh);
        }
        ORBX_WAVE_SYNC();
        // ---- D ----
        const int thr = __ballot(any_hi) != 0ull ? iniTh : minTh;
        for (int i0 = 0; i0 < total; i0 += kFastThreads) {
            const int i = i0 + lane;
            int flag = 0, p = 0, s = 0;
            if (i < total) {
                const int e = list[i];
                p = e & kListPos;
                s = sc[p];
                flag = (e >> 15) && s >= thr;
            }
            const unsigned long long bal = __ballot(flag);
            if (flag) {
                const int y1 = (int)((unsigned)mul24(p, (int)M) >> 20), x1 = p - mul24(y1, pitch);       // (y + 1, x + 1)
                out[base + __popcll(bal & lt)] = key_pack(ci.x0 + x1 - 1 - kBorder, ci.y0 + y1 - 1 - kBorder, s);
            }
            base += __popcll(bal);
        }
    } else {
        // ---- C', D' ----  the same over every interior pixel; the list region now holds one keep flag per score-tile byte
        uint8_t* kf = (uint8_t*)list;
        const unsigned Mi = (1u << 20) / (unsigned)iw + 1u;     // i / iw == (i * Mi) >> 20 exactly for i < 2^13
        const int npix = iw * ih;
        int any_hi = 0;
        for (int i = lane; i < npix; i += kFastThreads) {
            const int y = (int)((unsigned)mul24(i, (int)Mi) >> 20), x = i - mul24(y, iw);
            const int p = mul24(y + 1, pitch) + x + 1;
            const uint8_t* c = sc + p;
            const int s = c[0];
            const int m0 = imax(imax((int)c[-pitch - 1], (int)c[-pitch]), imax((int)c[-pitch + 1], (int)c[-1]));
            const int m1 = imax(imax((int)c[1], (int)c[pitch - 1]), imax((int)c[pitch], (int)c[pitch + 1]));
            const int keep = s > imax(m0, m1);
            kf[p] = (uint8_t)keep;
            any_hi |= (keep && s >= iniTh);
        }
        ORBX_WAVE_SYNC();
        const int thr = __ballot(any_hi) != 0ull ? iniTh : minTh;
        for (int i0 = 0; i0 < npix; i0 += kFastThreads) {
            const int i = i0 + lane;
            int flag = 0, x = 0, y = 0, s = 0;
            if (i < npix) {
                y = (int)((unsigned)mul24(i, (int)Mi) >> 20); x = i - mul24(y, iw);
                const int p = mul24(y + 1, pitch) + x + 1;
                s = sc[p];
                flag = kf[p] && s >= thr;
            }
            const unsigned long long bal = __ballot(flag);
            if (flag) out[base + __popcll(bal & lt)] = key_pack(ci.x0 + x - kBorder, ci.y0 + y - kBorder, s);
            base += __popcll(bal);
        }
    }
    if (lane == 0) cell_count[(size_t)b * ncells + cell] = base;
}

// ---------------------------------------------------------------------------------------------------
// 7x7 Gaussian blur, taps k[7] (symmetric; sum 256 or 257), REFLECT_101,
//   out = sat((sum_j k_j * (sum_i k_i * p) + 32768) >> 16).
// Streaming design: a thread owns 4 adjacent columns (one dword) and walks down a strip of kBlurRows rows with the
// last 7 horizontal sums in registers, so every input dword is fetched once per strip (+6 halo rows, L1/L2 hits) and
// every output is one coalesced dword store.  No LDS, no barriers.
// block (64,4): 256 columns x 4 strips.  grid (tiles over all levels, B): tile table in BlurTiles.
static_assert(kBlurRows % 2 == 0, "rows are produced in pairs");

__global__ void __launch_bounds__(256) k_blur(const LevelInfo* __restrict__ lv, int nlevels,
                                              const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                              size_t pyr_stride, BlurTaps taps, BlurTiles tiles) {
    int level = 0;
    for (int l = 1; l < nlevels; l++) if ((int)blockIdx.x >= tiles.begin[l]) level = l;
    const LevelInfo L = lv[level];
    const int b = (int)blockIdx.y;
    const int t = (int)blockIdx.x - tiles.begin[level];
    const int tcols = (L.w + 255) >> 8;
    const int ty = t / tcols, tx = t - ty * tcols;
    const int x0 = (tx * 64 + (int)threadIdx.x) * 4;
    const int ys = (ty * 4 + (int)threadIdx.y) * kBlurRows;
    if (x0 >= L.w || ys >= L.h) return;
    const uint8_t* src = pyr + (size_t)b * pyr_stride + L.off;
    uint8_t* dst = blur + (size_t)b * pyr_stride + L.off;
    const int k0 = taps.k[0], k1 = taps.k[1], k2 = taps.k[2], k3 = taps.k[3];
    // Loop-invariant REFLECT_101 column mapping: the 10 input columns x0-3..x0+6 are gathered from the 12-byte window
    // {l = x0-4.., c = x0.., r = x0+4..} by three byte-permutes whose selectors are computed once per thread, so border
    // lanes cost the same as interior lanes (no divergence).  Columns that only feed outputs >= w are don't-cares.
    uint32_t sel[3]; int base[3];
#pragma unroll
    for (int g = 0; g < 3; g++) {
        int idx[4], mn = 11;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int col = x0 - 3 + 4 * g + j;
            if (col < 0) col = -col;
            if (col >= L.w) col = 2 * L.w - 2 - col;
            idx[j] = imin(imax(col - (x0 - 4), 0), 11);
            if (g * 4 + j < 10) mn = imin(mn, idx[j]);
        }
        base[g] = imin(mn >> 2, 1);
        uint32_t sgl = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) sgl |= (uint32_t)imin(imax(idx[j] - 4 * base[g], 0), 7) << (8 * j);
        sel[g] = sgl;
    }
    // Horizontal pass: H_j = sum_i k_i p[j+i], j = 0..3, on the three byte windows P0 = p[0..3], P1 = p[4..7], P2 = p[8..9]: instead of
    // shifting the data to each j (v_alignbyte) the taps are shifted - ten v_dot4_u32_u8 with constant tap words; H <= 65535.
    // Vertical pass: the H of two consecutive input rows share a register (lo/hi 16 bits), so a 7-row window is four
    // v_dot2_u32_u16 with the taps paired to match the window's parity; the rounding constant is the accumulator's start value.
    const uint32_t uk0 = (uint32_t)k0, uk1 = (uint32_t)k1, uk2 = (uint32_t)k2, uk3 = (uint32_t)k3;
    const uint32_t T00 = uk0 | (uk1 << 8) | (uk2 << 16) | (uk3 << 24), T01 = uk2 | (uk1 << 8) | (uk0 << 16);                     // j = 0
    const uint32_t T10 = (uk0 << 8) | (uk1 << 16) | (uk2 << 24), T11 = uk3 | (uk2 << 8) | (uk1 << 16) | (uk0 << 24);             // j = 1
    const uint32_t T20 = (uk0 << 16) | (uk1 << 24), T21 = uk2 | (uk3 << 8) | (uk2 << 16) | (uk1 << 24), T22 = uk0;               // j = 2
    const uint32_t T30 = uk0 << 24, T31 = uk1 | (uk2 << 8) | (uk3 << 16) | (uk2 << 24), T32 = uk1 | (uk0 << 8);                  // j = 3
    const uint32_t Ke0 = (uint32_t)k0 | ((uint32_t)k1 << 16), Ke1 = (uint32_t)k2 | ((uint32_t)k3 << 16), Ke2 = (uint32_t)k2 | ((uint32_t)k1 << 16), Ke3 = (uint32_t)k0;
    const uint32_t Ko0 = (uint32_t)k0 << 16, Ko1 = (uint32_t)k1 | ((uint32_t)k2 << 16), Ko2 = (uint32_t)k3 | ((uint32_t)k2 << 16), Ko3 = (uint32_t)k1 | ((uint32_t)k0 << 16);
    // taps that sum to 256 cannot exceed 255 after the final shift ((255 * 65536 + 32768) >> 16 = 255): the four result bytes are then
    // cut out of the accumulators with two byte-permutes instead of shift + clamp + insert per output (wave-uniform choice)
    const bool exact256 = 2 * (k0 + k1 + k2) + k3 <= 256;
    uint32_t Q[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) { Q[i][0] = Q[i][1] = Q[i][2] = Q[i][3] = 0u; }
#pragma unroll
    for (int m = 0; m < (kBlurRows + 6) / 2; m++) {
        const int yo = ys + 2 * (m - 3);                // first of the two output rows completed by this pair of input rows
        if (m >= 3 && yo >= L.h) break;
        uint32_t Hr[2][4];
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            int y = ys - 3 + 2 * m + sub;
            if (y < 0) y = -y;
            if (y >= L.h) y = 2 * L.h - 2 - y;
            y = imax(y, 0);
            const uint8_t* row = src + (uint32_t)(mul24(y, L.pitch) + x0);      // uniform base + 32-bit offset: no 64-bit multiply-add per row
            const uint32_t c = *(const uint32_t*)row;
            const uint32_t l = x0 > 0 ? *(const uint32_t*)(row - 4) : 0u;
            const uint32_t r = x0 + 4 < L.pitch ? *(const uint32_t*)(row + 4) : 0u;
            const uint32_t P0 = byte_perm(base[0] ? r : c, base[0] ? c : l, sel[0]);      // input columns x0-3 .. x0
            const uint32_t P1 = byte_perm(base[1] ? r : c, base[1] ? c : l, sel[1]);      //               x0+1 .. x0+4
            const uint32_t P2 = byte_perm(base[2] ? r : c, base[2] ? c : l, sel[2]);      //               x0+5, x0+6, (unused)
            Hr[sub][0] = dot4_u8(P0, T00, dot4_u8(P1, T01, 0u));
            Hr[sub][1] = dot4_u8(P0, T10, dot4_u8(P1, T11, 0u));
            Hr[sub][2] = dot4_u8(P0, T20, dot4_u8(P1, T21, dot4_u8(P2, T22, 0u)));
            Hr[sub][3] = dot4_u8(P0, T30, dot4_u8(P1, T31, dot4_u8(P2, T32, 0u)));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { Q[0][j] = Q[1][j]; Q[1][j] = Q[2][j]; Q[2][j] = Q[3][j]; Q[3][j] = Hr[0][j] | (Hr[1][j] << 16); }
        if (m >= 3) {
            uint32_t ae[4], ao[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                ae[j] = dot2_u16(Q[0][j], Ke0, dot2_u16(Q[1][j], Ke1, dot2_u16(Q[2][j], Ke2, dot2_u16(Q[3][j], Ke3, 32768u))));
                ao[j] = dot2_u16(Q[0][j], Ko0, dot2_u16(Q[1][j], Ko1, dot2_u16(Q[2][j], Ko2, dot2_u16(Q[3][j], Ko3, 32768u))));
            }
            uint32_t oe = 0, oo = 0;
            if (exact256) {                            // byte 2 of each accumulator is the output
                oe = byte_perm(ae[1], ae[0], 0x0c0c0602u) | byte_perm(ae[3], ae[2], 0x06020c0cu);
                oo = byte_perm(ao[1], ao[0], 0x0c0c0602u) | byte_perm(ao[3], ao[2], 0x06020c0cu);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t ve = ae[j] >> 16, vo = ao[j] >> 16;
                    ve = ve > 255u ? 255u : ve; vo = vo > 255u ? 255u : vo;
                    oe |= ve << (8 * j); oo |= vo << (8 * j);
                }
            }
            const uint32_t oofs = (uint32_t)(mul24(yo, L.pitch) + x0);
            *(uint32_t*)(dst + oofs) = oe;
            if (yo + 1 < L.h) *(uint32_t*)(dst + (oofs + (uint32_t)L.pitch)) = oo;
        }
    }
}

}  // namespace orbx
