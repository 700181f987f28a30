// blur_body.h — the 7x7 Gaussian blur of one strip of one tile (one wave), shared by k_blur (k_image.hip) and by the small-batch launch that runs the
// blur beside the FAST cells (k_fast_cells_blur, k_fast.hip).
#pragma once
#include "orbx_types.h"
#include "orbx_block.h"
#include "orbx_kernels.h"
#include "orbx_simd.h"

namespace orbx {

// ---------------------------------------------------------------------------------------------------
// 7x7 Gaussian blur, taps k[7] (symmetric; sum 256 or 257), REFLECT_101,
//   out = sat((sum_j k_j * (sum_i k_i * p) + 32768) >> 16).
// Streaming design: a thread owns 4 adjacent columns (one dword) and walks down a strip of kRows rows with the
// last 7 horizontal sums in registers, so every input dword is fetched once per strip (+6 halo rows, L1/L2 hits) and
// every output is one coalesced dword store.  No LDS, no barriers.
// A tile is 256 columns x 4 strips; one wave owns one strip (tile = index into the tile table in BlurTiles, strip 0..3, lane 0..63).
static_assert(kBlurRows % 2 == 0 && kBlurRowsLarge % 2 == 0, "rows are produced in pairs");

template <int kRows>
__device__ __forceinline__ void blur_strip(const LevelInfo* __restrict__ lv, int nlevels, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                           size_t pyr_stride, const BlurTaps& taps, const BlurTiles& tiles, int tile, int strip, int lane, int b) {
    int level = 0;
    for (int l = 1; l < nlevels; l++) if (tile >= tiles.begin[l]) level = l;
    const LevelInfo L = lv[level];
    const int t = tile - tiles.begin[level];
    const int tcols = (L.w + 255) >> 8;
    const int ty = t / tcols, tx = t - ty * tcols;
    const int x0 = (tx * 64 + lane) * 4;
    const int ys = ORBX_UNIFORM((ty * 4 + strip) * kRows);      // one strip per wave: the row bookkeeping (reflection, row offsets) is scalar
    if (x0 >= L.w || ys >= L.h) return;
    const BufRsrc src = buf_make(pyr + (size_t)b * pyr_stride + L.off);
    const BufRsrc dst = buf_make(blur + (size_t)b * pyr_stride + L.off);
    const int k0 = taps.k[0], k1 = taps.k[1], k2 = taps.k[2], k3 = taps.k[3];
    // Loop-invariant REFLECT_101 column mapping: the 10 input columns x0-3..x0+6 are gathered from three dwords L, C, R of the row by three
    // byte-permutes over the fixed register pairs (C,L), (R,C), (R,C).  C is the thread's own dword; L is the dword to its left (the own one
    // again in the first column, whose reflected columns all lie in C); R is the dword to its right - or, in the last dwords of a row
    // (w - x0 <= 4), where every column right of the image reflects to the left, the dword to the LEFT (w - x0 = 4 needs neither).  The
    // dword offsets and the selectors are per-thread constants, so border lanes run the same instructions as interior lanes and no load is
    // predicated.  Columns that only feed outputs >= w are don't-cares.
    const int oL = x0 > 0 ? -4 : 0;
    const int oR = L.w - x0 > 4 ? 4 : oL;
    uint32_t sel[3];
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const int hi0 = g == 0 ? x0 : x0 + oR, lo0 = g == 0 ? x0 + oL : x0;     // first columns of the pair's high / low dword
        uint32_t sgl = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int col = x0 - 3 + 4 * g + j;
            if (col < 0) col = -col;
            if (col >= L.w) col = 2 * L.w - 2 - col;
            const int dl = col - lo0, dh = col - hi0;
            sgl |= ((unsigned)dl < 4u ? (uint32_t)dl : (unsigned)dh < 4u ? (uint32_t)(4 + dh) : 0u) << (8 * j);
        }
        sel[g] = sgl;
    }
    const int xl = x0 + oL, xr = x0 + oR;
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
    for (int m = 0; m < (kRows + 6) / 2; m++) {
        const int yo = ys + 2 * (m - 3);                // first of the two output rows completed by this pair of input rows
        if (m >= 3 && yo >= L.h) break;
        uint32_t Hr[2][4];
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            int y = ys - 3 + 2 * m + sub;
            if (y < 0) y = -y;
            if (y >= L.h) y = 2 * L.h - 2 - y;
            y = imax(y, 0);
            const uint32_t ro = (uint32_t)(y * L.pitch);                        // scalar row offset + per-thread column offsets: no vector address arithmetic
            const uint32_t c = buf_load_u32(src, (uint32_t)x0, ro);
            const uint32_t l = buf_load_u32(src, (uint32_t)xl, ro);
            const uint32_t r = buf_load_u32(src, (uint32_t)xr, ro);
            const uint32_t P0 = byte_perm(c, l, sel[0]);      // input columns x0-3 .. x0
            const uint32_t P1 = byte_perm(r, c, sel[1]);      //               x0+1 .. x0+4
            const uint32_t P2 = byte_perm(r, c, sel[2]);      //               x0+5, x0+6, (unused)
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
            const uint32_t oo_ = (uint32_t)(yo * L.pitch);
            buf_store_u32(oe, dst, (uint32_t)x0, oo_);
            if (yo + 1 < L.h) buf_store_u32(oo, dst, (uint32_t)x0, oo_ + (uint32_t)L.pitch);
        }
    }
}

}  // namespace orbx
