// k_image.hip — the image-space (streaming) kernels of the ORB extractor for gfx950:
//   k_resize      pyramid level l from level l-1   (reference ComputePyramid, src/ORBextractor.cc:1687-1738,
//                                                    arithmetic of cv::resize INTER_LINEAR 8U)
//   k_fast_cells  per-cell FAST-9/16 score + cell-local 3x3 NMS + per-cell threshold choice + ordered
//                 compaction               (reference ComputeKeyPointsOctTree, :1061-1166, cv::FAST)
//   k_blur        7x7 sigma=2 Gaussian, REFLECT_101, 8-bit fixed point        (:1629-1637, cv::GaussianBlur)
// All are HBM/LDS-bound integer kernels: no MFMA.  One launch covers every image of the batch.
#include "orbx_types.h"
#include "orbx_block.h"

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
#pragma unroll
    for (int k = 0; k < 4; k++) if (x0 + k < D.w) out |= (uint32_t)src[x0 + k] << (8 * k);
    *(uint32_t*)(pyr + (size_t)b * pyr_stride + D.off + (size_t)y * D.pitch + x0) = out;
}

// ---------------------------------------------------------------------------------------------------
// Pyramid: bilinear 11-bit fixed point, 4 output pixels per thread, one dword store.
// block (64,4): 256 output columns x 4 rows.  grid (ceil(pitch/256), ceil(h/4), B)
__global__ void __launch_bounds__(256) k_resize(const LevelInfo* __restrict__ lv, int level,
                                                const ResizeTap* __restrict__ xtab,
                                                const ResizeTap* __restrict__ ytab,
                                                uint8_t* __restrict__ pyr, size_t pyr_stride) {
    const LevelInfo D = lv[level];
    const LevelInfo S = lv[level - 1];
    const int b = (int)blockIdx.z;
    const int dy = (int)(blockIdx.y * 4 + threadIdx.y);
    const int dx0 = (int)(blockIdx.x * 64 + threadIdx.x) * 4;
    if (dy >= D.h || dx0 >= D.pitch) return;
    const uint8_t* src = pyr + (size_t)b * pyr_stride + S.off;
    uint8_t* dst = pyr + (size_t)b * pyr_stride + D.off;
    const ResizeTap ty = ytab[D.ytab_off + dy];
    const int sy0 = imin(imax(ty.ofs, 0), S.h - 1), sy1 = imin(imax(ty.ofs + 1, 0), S.h - 1);
    const int b0 = (int)(int16_t)(ty.w & 0xFFFF), b1 = ty.w >> 16;
    const uint8_t* S0 = src + (size_t)sy0 * S.pitch;
    const uint8_t* S1 = src + (size_t)sy1 * S.pitch;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dx = dx0 + k;
        if (dx < D.w) {
            const ResizeTap tx = xtab[D.xtab_off + dx];
            const int sx = tx.ofs, sx1 = imin(sx + 1, S.w - 1);
            const int a0 = (int)(int16_t)(tx.w & 0xFFFF), a1 = tx.w >> 16;
            const int r0 = S0[sx] * a0 + S0[sx1] * a1;
            const int r1 = S1[sx] * a0 + S1[sx1] * a1;
            int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            v = imin(imax(v, 0), 255);
            out |= (uint32_t)v << (8 * k);
        }
    }
    *(uint32_t*)(dst + (size_t)dy * D.pitch + dx0) = out;
}

// ---------------------------------------------------------------------------------------------------
// FAST-9/16.  ring offsets (dx,dy), k = 0..15, as in OpenCV: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)
// (0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
// Returns OpenCV's cornerScore (largest threshold for which the pixel is still a corner) when the pixel
// is a corner at threshold t0, else 0.   score = max over the 16 nine-arcs of min |v - ring| , minus 1.
__device__ __forceinline__ int fast_score(const uint8_t* c, int wp, int t0) {
    const int v = c[0];
    int d[16];
    d[0] = v - c[3 * wp];          d[1] = v - c[3 * wp + 1];    d[2] = v - c[2 * wp + 2];    d[3] = v - c[wp + 3];
    d[4] = v - c[3];               d[5] = v - c[-wp + 3];       d[6] = v - c[-2 * wp + 2];   d[7] = v - c[-3 * wp + 1];
    d[8] = v - c[-3 * wp];         d[9] = v - c[-3 * wp - 1];   d[10] = v - c[-2 * wp - 2];  d[11] = v - c[-wp - 3];
    d[12] = v - c[-3];             d[13] = v - c[wp - 3];       d[14] = v - c[2 * wp - 2];   d[15] = v - c[3 * wp - 1];
    // exact quick rejection: a 9-arc contains one pixel of every opposite pair (k, k+8)
    bool dark = true, bright = true;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        dark = dark && (d[k] > t0 || d[k + 8] > t0);
        bright = bright && (d[k] < -t0 || d[k + 8] < -t0);
    }
    if (!dark && !bright) return 0;
    // sliding 9-window min (dark arcs) and max (bright arcs) over the circular ring by doubling
    int mn2[16], mx2[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn2[k] = imin(d[k], d[(k + 1) & 15]); mx2[k] = imax(d[k], d[(k + 1) & 15]); }
    int mn4[16], mx4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn4[k] = imin(mn2[k], mn2[(k + 2) & 15]); mx4[k] = imax(mx2[k], mx2[(k + 2) & 15]); }
    int Md = -255, Mb = -255;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int mn9 = imin(imin(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int mx9 = imax(imax(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
        Md = imax(Md, mn9);
        Mb = imax(Mb, -mx9);
    }
    const int m = imax(Md, Mb);
    return m > t0 ? m - 1 : 0;
}

// One workgroup per (cell, image).  LDS: window tile | score tile | nms tile.
// slots: per-cell candidate lists in the reference order (row-major inside the cell);
// cell_count[b*ncells + cell] = number of candidates kept for the cell.
__global__ void __launch_bounds__(256) k_fast_cells(const LevelInfo* __restrict__ lv,
                                                    const CellInfo* __restrict__ cells, int ncells,
                                                    const uint8_t* __restrict__ pyr, size_t pyr_stride,
                                                    int iniTh, int minTh,
                                                    uint32_t* __restrict__ slots, size_t slots_stride,
                                                    int* __restrict__ cell_count, int tile_bytes, int inner_bytes) {
    ORBX_DYN_SMEM(smem);
    __shared__ int s_flags[2];
    __shared__ int s_wave[4];
    const int cell = (int)blockIdx.x, b = (int)blockIdx.y;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const CellInfo ci = cells[cell];
    const LevelInfo L = lv[ci.level];
    const int iw = ci.x1 - ci.x0, ih = ci.y1 - ci.y0;
    if (iw <= 0 || ih <= 0) {
        if (tid == 0) cell_count[(size_t)b * ncells + cell] = 0;
        return;
    }
    const int ww = iw + 6, wh = ih + 6, wp = (ww + 3) & ~3;
    uint8_t* tile = smem;
    uint8_t* sc = smem + tile_bytes;
    uint8_t* nm = sc + inner_bytes;
    const uint8_t* img = pyr + (size_t)b * pyr_stride + L.off;
    for (int i = tid; i < wh * ww; i += 256) {
        const int r = i / ww, c = i - r * ww;
        tile[r * wp + c] = img[(size_t)(ci.y0 - 3 + r) * L.pitch + (ci.x0 - 3 + c)];
    }
    if (tid == 0) { s_flags[0] = 0; }
    __syncthreads();
    const int t0 = imin(iniTh, minTh);
    const int npx = iw * ih;
    for (int p = tid; p < npx; p += 256) {
        const int y = p / iw, x = p - y * iw;
        sc[p] = (uint8_t)fast_score(tile + (y + 3) * wp + (x + 3), wp, t0);
    }
    __syncthreads();
    // cell-local strict 3x3 non-max suppression (neighbours outside the interior count as 0)
    int any_hi = 0;
    for (int p = tid; p < npx; p += 256) {
        const int y = p / iw, x = p - y * iw;
        const int s = sc[p];
        int keep = 0;
        if (s > 0) {
            keep = 1;
#pragma unroll
            for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                for (int dx = -1; dx <= 1; dx++) {
                    if (dx == 0 && dy == 0) continue;
                    const int xx = x + dx, yy = y + dy;
                    const int q = (xx >= 0 && xx < iw && yy >= 0 && yy < ih) ? (int)sc[yy * iw + xx] : 0;
                    keep &= (s > q);
                }
        }
        nm[p] = keep ? (uint8_t)s : (uint8_t)0;
        any_hi |= (keep && s >= iniTh);
    }
    if (any_hi) atomicOr(&s_flags[0], 1);
    __syncthreads();
    // threshold choice of the reference: FAST at iniTh; if that yields nothing, FAST at minTh (:1135-1148)
    const int thr = s_flags[0] ? iniTh : minTh;
    uint32_t* out = slots + (size_t)b * slots_stride + ci.slot_off;
    int base = 0;
    for (int p0 = 0; p0 < npx; p0 += 256) {
        const int p = p0 + tid;
        int s = 0;
        if (p < npx) s = nm[p];
        const int flag = (s > 0 && s >= thr);
        const unsigned long long bal = __ballot(flag);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < 4; w++) { const int c = s_wave[w]; if (w < wave) wbase += c; tot += c; }
        if (flag) {
            const int y = p / iw, x = p - y * iw;
            const int pos = base + wbase + __popcll(bal & ((1ull << lane) - 1ull));
            out[pos] = key_pack(ci.x0 + x - kBorder, ci.y0 + y - kBorder, s);
        }
        base += tot;
        __syncthreads();
    }
    if (tid == 0) cell_count[(size_t)b * ncells + cell] = base;
}

// ---------------------------------------------------------------------------------------------------
// 7x7 Gaussian blur, taps k[7] (sum 256 or 257), REFLECT_101, out = sat((sum_j k_j * (sum_i k_i*p) + 32768) >> 16).
// block (64,4); tile 64 x 16 outputs.  grid (ceil(w0/64), ceil(h0/16), B*nlevels); blocks outside a level exit.
__global__ void __launch_bounds__(256) k_blur(const LevelInfo* __restrict__ lv, int nlevels,
                                              const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                              size_t pyr_stride, BlurTaps taps) {
    __shared__ uint8_t s_in[22][72];
    __shared__ uint16_t s_h[22][64];
    const int level = (int)(blockIdx.z % (unsigned)nlevels), b = (int)(blockIdx.z / (unsigned)nlevels);
    const LevelInfo L = lv[level];
    const int x0 = (int)blockIdx.x * 64, y0 = (int)blockIdx.y * 16;
    if (x0 >= L.w || y0 >= L.h) return;
    const int tid = (int)(threadIdx.y * 64 + threadIdx.x);
    const uint8_t* src = pyr + (size_t)b * pyr_stride + L.off;
    uint8_t* dst = blur + (size_t)b * pyr_stride + L.off;
    for (int i = tid; i < 22 * 70; i += 256) {
        const int r = i / 70, c = i - r * 70;
        int yy = y0 + r - 3, xx = x0 + c - 3;
        // reflect101; coordinates far outside (tile overhang past the image) are clamped after reflection
        if (yy < 0) yy = -yy;
        if (yy >= L.h) yy = 2 * L.h - 2 - yy;
        if (xx < 0) xx = -xx;
        if (xx >= L.w) xx = 2 * L.w - 2 - xx;
        yy = imin(imax(yy, 0), L.h - 1); xx = imin(imax(xx, 0), L.w - 1);
        s_in[r][c] = src[(size_t)yy * L.pitch + xx];
    }
    __syncthreads();
    for (int i = tid; i < 22 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        int s = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) s += taps.k[k] * s_in[r][c + k];
        s_h[r][c] = (uint16_t)s;
    }
    __syncthreads();
    const int c = (int)threadIdx.x;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int r = (int)threadIdx.y * 4 + rr;
        unsigned s = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) s += (unsigned)taps.k[k] * (unsigned)s_h[r + k][c];
        unsigned v = (s + 32768u) >> 16;
        v = v > 255u ? 255u : v;
        if (x0 + c < L.w && y0 + r < L.h) dst[(size_t)(y0 + r) * L.pitch + x0 + c] = (uint8_t)v;
    }
}

}  // namespace orbx
