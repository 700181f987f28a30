// k_image.hip — the image-space (streaming) kernels of the ORB extractor for gfx950:
//   k_resize      pyramid level l from level l-1   (reference ComputePyramid, src/ORBextractor.cc:1687-1738,
//                                                    arithmetic of cv::resize INTER_LINEAR 8U)
//   k_blur        7x7 sigma=2 Gaussian, REFLECT_101, 8-bit fixed point        (:1629-1637, cv::GaussianBlur)
// All are HBM/LDS-bound integer kernels: no MFMA.  One launch covers every image of the batch.
#include "orbx_types.h"
#include "orbx_block.h"
#include "orbx_kernels.h"
#include "orbx_simd.h"
#include "blur_body.h"

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
// Pyramid, small batches: every level in ONE launch.  A chain of seven dependent launches costs 4.5 us each on this part whatever the
// work (tools/sync_latency_probe.hip), which is all of the pyramid's 28 us for a single stereo pair.  Level l depends on level l - 1
// only locally (two taps per axis), so a workgroup that is given a tile of the TOP level can derive everything below it from a window of
// level 0 alone: the window goes to LDS, each level's region is interpolated from the region of the level before (both in LDS, ping-pong),
// and every level is partitioned among the tiles (PyrSpan::o0 / o1, built by the host from the tap tables: a tile owns at level l what
// starts at the left tap of the first pixel it owns at level l + 1), so each workgroup writes the part of each level it owns.  Regions
// overlap by the taps' reach (about a third more pixels than a level holds); the arithmetic is k_resize's.  Columns are handled as
// dwords: regions start and end on multiples of 4 and a dword that straddles two owners is written by both, with the same bytes.
// block kPyrThreads; grid (tiles_x * tiles_y, B); dynamic LDS = bufA + bufB + 8 * (region widths + heights of all levels) + 48 * nlevels.
__global__ void __launch_bounds__(kPyrThreads) k_pyramid_fused(const LevelInfo* __restrict__ lv, int nlevels, const ResizeTap* __restrict__ xtab,
                                                       const ResizeTap* __restrict__ ytab, const PyrSpan* __restrict__ xspan,
                                                       const PyrSpan* __restrict__ yspan, int ntx, uint8_t* __restrict__ pyr, size_t pyr_stride,
                                                       int buf_a_bytes, int buf_b_bytes, PyrTapOffsets toff) {
    ORBX_DYN_SMEM(smem);
    const int tid = (int)threadIdx.x, b = (int)blockIdx.y;
    const int ty = (int)blockIdx.x / ntx, tx = (int)blockIdx.x - ty * ntx;
    // taps of every level, rebased to the source regions: per region column (row) the two source offsets (lo | hi << 16) and the weight pair
    uint32_t* taps = (uint32_t*)(smem + buf_a_bytes + buf_b_bytes);
    PyrSpan* spx = (PyrSpan*)(taps + 2 * toff.total);                        // this tile's spans, all levels
    PyrSpan* spy = spx + nlevels;
    int* lvi = (int*)(spy + nlevels);                                        // per level: w, h, pitch, off, xtab_off, ytab_off (6 ints)
    int* tox = lvi + 6 * nlevels;                                            // toff.x / toff.y, indexed per lane below
    int* toy = tox + nlevels;
    uint8_t* img = pyr + (size_t)b * pyr_stride;
    // Everything that comes from global memory is fetched up front, in two rounds of independent loads - spans and level records, then the
    // level-0 window and the taps of all levels - so that the level loop below only touches LDS (a memory latency per level and per table
    // was most of the first version's 18 us).
    if (tid < nlevels) spx[tid] = xspan[tx * nlevels + tid];
    else if (tid < 2 * nlevels) spy[tid - nlevels] = yspan[ty * nlevels + (tid - nlevels)];
    else if (tid >= 64 && tid < 64 + nlevels) {
        const LevelInfo L = lv[tid - 64];
        int* q = lvi + 6 * (tid - 64);
        q[0] = L.w; q[1] = L.h; q[2] = L.pitch; q[3] = L.off; q[4] = L.xtab_off; q[5] = L.ytab_off;
    } else if (tid >= 128 && tid < 128 + nlevels) { tox[tid - 128] = toff.x[tid - 128]; toy[tid - 128] = toff.y[tid - 128]; }
    __syncthreads();
    {
        // taps: entry e of the concatenated list [level 1 columns | level 1 rows | level 2 columns | ...]; four entries per thread in flight
        constexpr int kU = 4;
        for (int e0 = tid; e0 < toff.total; e0 += kPyrThreads * kU) {
            ResizeTap t[kU]; int lvl[kU], idx[kU]; bool col[kU];
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const int e = e0 + kPyrThreads * u;
                lvl[u] = 0;
                if (e < toff.total) {
                    int l = 1;
                    while (l + 1 < nlevels && e >= tox[l + 1]) l++;
                    col[u] = e < toy[l];
                    idx[u] = e - (col[u] ? tox[l] : toy[l]);
                    const PyrSpan d = col[u] ? spx[l] : spy[l];
                    if (idx[u] < d.b - d.a) {                                // (the list is sized for the widest tile)
                        lvl[u] = l;
                        t[u] = col[u] ? xtab[lvi[6 * l + 4] + imin(d.a + idx[u], lvi[6 * l] - 1)] : ytab[lvi[6 * l + 5] + d.a + idx[u]];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const int l = lvl[u];
                if (l > 0) {
                    const int e = e0 + kPyrThreads * u;
                    const int lim = (col[u] ? lvi[6 * (l - 1)] : lvi[6 * (l - 1) + 1]) - 1, org = col[u] ? spx[l - 1].a : spy[l - 1].a;
                    const int o0 = imin(imax(t[u].ofs, 0), lim) - org, o1 = imin(imax(t[u].ofs + 1, 0), lim) - org;
                    taps[2 * e] = (uint32_t)o0 | ((uint32_t)o1 << 16);
                    taps[2 * e + 1] = (uint32_t)t[u].w;
                }
            }
        }
        const PyrSpan sx = spx[0], sy = spy[0];
        const int pitch0 = lvi[2];
        const int ndw = (sx.b - sx.a) >> 2, n = ndw * (sy.b - sy.a);
        const unsigned Mc = (1u << 20) / (unsigned)ndw + 1u;                // i / ndw == (i * Mc) >> 20 exactly for i < 2^13 (the host checks the region sizes)
        const uint8_t* src = img + lvi[3] + mul24(sy.a, pitch0) + sx.a;
        for (int i0 = tid; i0 < n; i0 += kPyrThreads * kU) {
            uint32_t v[kU];
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const int i = i0 + kPyrThreads * u;
                if (i < n) {
                    const int r = (int)((unsigned)mul24(i, (int)Mc) >> 20), c = i - mul24(r, ndw);
                    v[u] = *(const uint32_t*)(src + mul24(r, pitch0) + 4 * c);
                }
            }
#pragma unroll
            for (int u = 0; u < kU; u++) { const int i = i0 + kPyrThreads * u; if (i < n) *(uint32_t*)(smem + 4 * i) = v[u]; }
        }
    }
    __syncthreads();
    for (int l = 1; l < nlevels; l++) {
        const int Dw = lvi[6 * l], Dpitch = lvi[6 * l + 2];
        const PyrSpan dx = spx[l], dy = spy[l];
        const int sw = spx[l - 1].b - spx[l - 1].a;                          // pitch of the source region
        const int dw = dx.b - dx.a, dh = dy.b - dy.a;
        const uint32_t* tapx = taps + 2 * tox[l];
        const uint32_t* tapy = taps + 2 * toy[l];
        // (offsets into the one LDS array, not pointers picked from an array: the compiler keeps LDS addressing only that way)
        const int so = (l & 1) ? 0 : buf_a_bytes, dofs = (l & 1) ? buf_a_bytes : 0;
        uint8_t* dst = img + lvi[6 * l + 3];
        const int ndw = dw >> 2, n = ndw * dh;
        const unsigned Mc = (1u << 20) / (unsigned)ndw + 1u;
        for (int i = tid; i < n; i += kPyrThreads) {
            const int r = (int)((unsigned)mul24(i, (int)Mc) >> 20), c = i - mul24(r, ndw);
            const uint32_t ty0 = tapy[2 * r], tyw = tapy[2 * r + 1];
            const int s0 = so + mul24((int)(ty0 & 0xFFFFu), sw), s1 = so + mul24((int)(ty0 >> 16), sw);
            const int b0 = (int)(int16_t)(tyw & 0xFFFFu), b1 = (int)tyw >> 16;
            // the staged taps of columns beyond the level's width repeat the last column's, so all four pixels are computed without branches
            // (their sixteen byte reads go out together) and the padding bytes are cleared afterwards
            uint32_t tx0[4], txw[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { tx0[k] = tapx[2 * (4 * c + k)]; txw[k] = tapx[2 * (4 * c + k) + 1]; }
            int p00[4], p01[4], p10[4], p11[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int o0 = (int)(tx0[k] & 0xFFFFu), o1 = (int)(tx0[k] >> 16);
                p00[k] = smem[s0 + o0]; p01[k] = smem[s0 + o1]; p10[k] = smem[s1 + o0]; p11[k] = smem[s1 + o1];
            }
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int a0 = (int)(int16_t)(txw[k] & 0xFFFFu), a1 = (int)txw[k] >> 16;
                const int h0 = mul24(p00[k], a0) + mul24(p01[k], a1);
                const int h1 = mul24(p10[k], a0) + mul24(p11[k], a1);
                int v = ((mul24_forced(b0, h0 >> 4) >> 16) + (mul24_forced(b1, h1 >> 4) >> 16) + 2) >> 2;
                v = imin(imax(v, 0), 255);
                out |= (uint32_t)v << (8 * k);
            }
            const int x = dx.a + 4 * c, y = dy.a + r;
            const int nvalid = Dw - x;                                      // columns of this dword inside the level (>= 1)
            if (nvalid < 4) out &= 0xFFFFFFFFu >> (8 * (4 - nvalid));
            *(uint32_t*)(smem + dofs + 4 * i) = out;
            if (y >= dy.o0 && y < dy.o1 && x + 3 >= dx.o0 && x < dx.o1) *(uint32_t*)(dst + mul24(y, Dpitch) + x) = out;
        }
        __syncthreads();                                                    // the region of level l is complete
    }
}

// ---------------------------------------------------------------------------------------------------
// Pyramid, streaming form (the common case: scale factor <= 2).  A thread owns 4 adjacent output columns and walks down kResizeStrip
// output rows.  Per source row it loads the 8 source bytes its columns need (one unaligned 8-byte load; neighbouring threads overlap in
// L1), cuts the two taps of every column out with one v_perm_b32 (as a u16 pair) and forms the horizontal sum with one v_dot2_u32_u16
// against the column's weight pair; consecutive output rows share a source row (5 times out of 6 at scale 1.2), so about 1.2 horizontal
// rows are computed per output row.  Vertical pass as in cv::resize: ((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2.
// No LDS, no barriers, ~15 VALU instructions per output pixel (the tile-staged k_resize above needs ~50).
// block (64, 4): 256 columns x 4 strips; grid (ceil(pitch / 256), ceil(h / (4 * strip_rows)), B).
__global__ void __launch_bounds__(256) k_resize_rows(const LevelInfo* __restrict__ lv, int level,
                                                     const ResizeTap* __restrict__ xtab, const ResizeTap* __restrict__ ytab,
                                                     uint8_t* __restrict__ pyr, size_t pyr_stride, int strip_rows) {
    const LevelInfo D = lv[level];
    const LevelInfo S = lv[level - 1];
    const int b = (int)blockIdx.z;
    // (lanes beyond the row pitch - the last workgroup of a row - redo its last dword and store the same bytes: every lane of a wave stays alive, because
    // lane i also carries the vertical taps of output row ys + i for the whole wave, below.  Until round 6 those lanes left the kernel, which was correct only
    // while a strip had at most 16 rows = the lanes a 64-byte pitch always fills: ORBX_RESIZE_STRIP = 32 read taps out of dead lanes.)
    const int dx0 = imin(((int)blockIdx.x * 64 + (int)threadIdx.x) * 4, D.pitch - 4);
#ifdef ORBX_EMU
    const int strip = (int)blockIdx.y * 4 + (int)threadIdx.y;
#else
    const int strip = __builtin_amdgcn_readfirstlane((int)blockIdx.y * 4 + (int)threadIdx.y);     // one strip per wave: row bookkeeping stays scalar
#endif
    const int ys = strip * strip_rows;
    if (ys >= D.h) return;
    const BufRsrc src = buf_make(pyr + (size_t)b * pyr_stride + S.off);
    const BufRsrc dst = buf_make(pyr + (size_t)b * pyr_stride + D.off);
    const ResizeTap* xt = xtab + D.xtab_off;
    const ResizeTap* yt = ytab + D.ytab_off;
    // per-column constants: byte selectors into the 8-byte source window that starts at the first column's left tap, and the weight pairs
    uint32_t sel[4], wgt[4];
    const int sx0 = xt[imin(dx0, D.w - 1)].ofs;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const ResizeTap tx = xt[imin(dx0 + k, D.w - 1)];
        const int i0 = tx.ofs - sx0, i1 = imin(tx.ofs + 1, S.w - 1) - sx0;              // 0..7 (checked on the host for every level that uses this kernel)
        sel[k] = (uint32_t)i0 | 0x0c00u | ((uint32_t)i1 << 16) | 0x0c000000u;           // bytes (tap0, 0, tap1, 0)
        wgt[k] = dx0 + k < D.w ? (uint32_t)tx.w : 0u;                                   // a0 | a1 << 16; columns in the row padding come out 0
    }
    // one (unaligned) 8-byte load: scalar row offset + the thread's first source column, no vector address arithmetic
    auto load_row = [&](int r) { return buf_load_u64(src, (uint32_t)sx0, (uint32_t)(imin(r, S.h - 1) * S.pitch)); };
    // The strip's source rows are walked once, in order, with the loads two rows ahead of their use; an output row is emitted when its
    // second source row arrives (its first one is the previous row, or the same row where cv::resize clamps at the image border).
    const int ye = imin(ys + strip_rows, D.h);
    int dy = ys;
    // the strip's vertical taps: lane i holds those of row ys + i, the row loop reads them back as wave-uniform scalars (no loads in the loop
    // besides the source rows)
    const ResizeTap tyl = yt[imin(ys + ((int)threadIdx.x & 63), D.h - 1)];                 // strip_rows <= 64
    ResizeTap ty; ty.ofs = ORBX_READLANE(tyl.ofs, 0); ty.w = ORBX_READLANE(tyl.w, 0);
    const int r_first = imin(imax(ty.ofs, 0), S.h - 1), r_last = imin(imax(ORBX_READLANE(tyl.ofs, ye - 1 - ys) + 1, 0), S.h - 1);
    // horizontal pass of one source row
    auto hrow = [&](const u32x2& w, uint32_t* H) {
#pragma unroll
        for (int k = 0; k < 4; k++) H[k] = dot2_u16(byte_perm(w.hi, w.lo, sel[k]), wgt[k], 0u) & ~15u;      // (S >> 4) << 4: the vertical pass multiplies by 2^12 * beta and keeps bits 32.. of the product
    };
    // the output rows whose second source row is r (Hc), the first one being the row before (Hp) or the same row
    auto emit = [&](int r, const uint32_t* Hp, const uint32_t* Hc) {
        while (dy < ye && imin(imax(ty.ofs + 1, 0), S.h - 1) == r) {
            const bool same = imin(imax(ty.ofs, 0), S.h - 1) == r;             // both taps on this row (border clamp)
            // cv::resize: ((beta0 * (S0 >> 4)) >> 16) + ((beta1 * (S1 >> 4)) >> 16) + 2) >> 2 with 0 <= beta <= 2048 and S <= 255 * 2048.  H holds (S >> 4) << 4 <
            // 2^19 and the weights are scaled by 2^12 (< 2^24, scalar registers): (beta * (S >> 4)) >> 16 is then bits 32.. of a 24 x 24-bit product, one
            // v_mul_hi_u32_u24 per term; the sum cannot exceed 4 * 255 + 3, so two sums share a register and one byte-permute cuts the four results out
            const uint32_t b0 = (uint32_t)(ty.w & 0xFFFF) << 12, b1 = (uint32_t)(ty.w >> 16) << 12;
            uint32_t s[4];
#pragma unroll
            for (int k = 0; k < 4; k++) s[k] = add3_u32(mulhi_u24_uniform(b0, same ? Hc[k] : Hp[k]), mulhi_u24_uniform(b1, Hc[k]), 2u);
            const uint32_t out = byte_perm((s[2] | (s[3] << 16)) >> 2, (s[0] | (s[1] << 16)) >> 2, 0x06040200u);
            buf_store_u32(out, dst, (uint32_t)dx0, (uint32_t)(dy * D.pitch));
            dy++;
            if (dy < ye) { ty.ofs = ORBX_READLANE(tyl.ofs, dy - ys); ty.w = ORBX_READLANE(tyl.w, dy - ys); }
        }
    };
    // three row registers and two rows of horizontal sums rotate by NAME through six copies of the loop body (no register moves)
    u32x2 w[3];
    uint32_t H[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    w[0] = load_row(r_first); w[1] = load_row(r_first + 1);
    int r = r_first;
#define ORBX_STEP(i) { if (r > r_last) break; w[((i) + 2) % 3] = load_row(r + 2); hrow(w[(i) % 3], H[(i) % 2]); emit(r, H[((i) + 1) % 2], H[(i) % 2]); r++; }
    for (;;) { ORBX_STEP(0) ORBX_STEP(1) ORBX_STEP(2) ORBX_STEP(3) ORBX_STEP(4) ORBX_STEP(5) }
#undef ORBX_STEP
}

// ---------------------------------------------------------------------------------------------------
// Self-test of the instruction wrappers of orbx_simd.h (orbx_debug_simd_selftest): out[op * n + i] = op(a[i], b[i], c[i]).  The CPU tests
// run the kernels on plain-C stand-ins of these instructions; this entry lets the tests compare instruction and stand-in with an independent
// definition, operand by operand.  ops: 0 mul24, 1 mul24_forced, 2 byte_perm, 3 align_byte, 4 dot4_u8, 5 dot2_u16, 6 pk_max3, 7 pk_min3,
// 8 pk_sub, 9 pk_xor(a, c), 10 wave_incl_scan, 11 wave_sum (both of a & 0xFFFF), 12 wave_min_u32(b), 13 sad4_u8, 14 __umul24,
// 15 / 16 wave_incl_scan of a 64-bit value (low / high word), 17 / 18 block_excl_scan_n of it + 2 x total, 19 wave_or_u32, 20 mulhi_u24, 21 add3_u32
// (kSimdSelftestOps in all).
__global__ void __launch_bounds__(256) k_simd_selftest(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ c, int n,
                                                       uint32_t* __restrict__ out) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= n) return;
    const uint32_t x = a[i], y = b[i], z = c[i];
    auto bits = [](pk2 v) { return (uint32_t)(uint16_t)pk_lo(v) | ((uint32_t)(uint16_t)pk_hi(v) << 16); };
    out[0 * (size_t)n + i] = (uint32_t)mul24((int)x, (int)y);
    out[1 * (size_t)n + i] = (uint32_t)mul24_forced((int)x, (int)y);
    out[2 * (size_t)n + i] = byte_perm(x, y, z);
    out[3 * (size_t)n + i] = align_byte(x, y, z);
    out[4 * (size_t)n + i] = dot4_u8(x, y, z);
    out[5 * (size_t)n + i] = dot2_u16(x, y, z);
    out[6 * (size_t)n + i] = bits(pk_max3(pk_make(x), pk_make(y), pk_make(z)));
    out[7 * (size_t)n + i] = bits(pk_min3(pk_make(x), pk_make(y), pk_make(z)));
    out[8 * (size_t)n + i] = bits(pk_sub(pk_make(x), pk_make(y)));
    out[9 * (size_t)n + i] = bits(pk_xor(pk_make(x), z));
    // wave primitives of orbx_block.h (on the GPU the inclusive scan of 32-bit integers is a DPP sequence): per wave of 64 consecutive elements
    // (n is a multiple of 256: every lane is active)
    out[10 * (size_t)n + i] = (uint32_t)wave_incl_scan<int>((int)(x & 0xFFFFu));
    out[11 * (size_t)n + i] = (uint32_t)wave_sum<int>((int)(x & 0xFFFFu));
    out[12 * (size_t)n + i] = wave_min_u32(y);
    out[13 * (size_t)n + i] = sad4_u8(x, y, z);
    out[14 * (size_t)n + i] = (uint32_t)__umul24(x, y);
    // 64-bit scans (three 20-bit fields, as k_quadtree packs them): over the wave, and exclusive over the four waves of this workgroup
    __shared__ unsigned long long s_scan[20];
    const unsigned long long f = (unsigned long long)(x & 0xFFFFFu) | ((unsigned long long)(y & 0xFFFFFu) << 20) | ((unsigned long long)(z & 0xFFFFFu) << 40);
    const unsigned long long ws = wave_incl_scan<unsigned long long>(f);
    out[15 * (size_t)n + i] = (uint32_t)ws; out[16 * (size_t)n + i] = (uint32_t)(ws >> 32);
    unsigned long long tot;
    const unsigned long long bs = block_excl_scan_n<unsigned long long>(f, &tot, s_scan, 4) + (tot << 1);      // (the total goes in as well)
    out[17 * (size_t)n + i] = (uint32_t)bs; out[18 * (size_t)n + i] = (uint32_t)(bs >> 32);
    out[19 * (size_t)n + i] = wave_or_u32((y & 7u) == 0u ? 1u << (x & 31u) : 0u);
    out[20 * (size_t)n + i] = mulhi_u24(x, y);
    out[21 * (size_t)n + i] = add3_u32(x, y, z);
}

// ---------------------------------------------------------------------------------------------------
// 7x7 Gaussian blur (blur_body.h).  block (64,4): 256 columns x 4 strips.  grid (tiles over all levels, B): tile table in BlurTiles.
__global__ void __launch_bounds__(256) k_blur(const LevelInfo* __restrict__ lv, int nlevels,
                                              const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                              size_t pyr_stride, BlurTaps taps, BlurTiles tiles) {
    blur_strip<kBlurRows>(lv, nlevels, pyr, blur, pyr_stride, taps, tiles, (int)blockIdx.x, (int)threadIdx.y, (int)threadIdx.x, (int)blockIdx.y);
}
// the same with strips of kBlurRowsLarge rows (large batches; the tile table counts tiles of 4 x kBlurRowsLarge rows)
__global__ void __launch_bounds__(256) k_blur_large(const LevelInfo* __restrict__ lv, int nlevels,
                                                    const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                                    size_t pyr_stride, BlurTaps taps, BlurTiles tiles) {
    blur_strip<kBlurRowsLarge>(lv, nlevels, pyr, blur, pyr_stride, taps, tiles, (int)blockIdx.x, (int)threadIdx.y, (int)threadIdx.x, (int)blockIdx.y);
}

}  // namespace orbx
