// k_fast.hip — k_fast_cells: per-cell FAST-9/16 score + cell-local 3x3 NMS + per-cell threshold choice + ordered compaction
// (reference ComputeKeyPointsOctTree, src/ORBextractor.cc:1061-1166, on top of cv::FAST(img, kps, th, nonmaxSuppression = true)).
//
// Per cell (one single-wave workgroup), F2 of SURVEY.md: result = {p : s(p) >= iniTh and p a strict 3x3 maximum} if that set is not empty,
// else the same with minTh, where s() = OpenCV's cornerScore (0 if the pixel is not a corner at the threshold of the run), neighbours
// outside the cell's detectable interior count as 0, and the output order is row-major.  The phases below run at t0 = iniTh; a cell that
// yields nothing runs them again at t0 = minTh (the reference's own order; rare on textured imagery).
//
// Instruction budget.  On gfx950 only the two-operand VOP2 integer forms (v_add/sub/and/or/xor/lshr/mov) issue in ~2.5 clk per wave64;
// every VOP3 form (v_perm, v_bitop3, v_lshl_or, v_bfe, v_mbcnt, all packed 16-bit ops, v_cmp) takes ~4.4 clk (profiles/r02/valu_survey.txt).
// The phases are built around that:
//   load  window tile (dword-aligned) in LDS
//   A     quick rejection of every interior pixel, 4 adjacent pixels per lane in one register (SWAR) in the "H form" (p >> 1) | 0x80 of
//         every byte (7-bit value + guard bit): one v_sub_u32 compares four ring pixels with four thresholds, opposite ring pairs combine
//         with v_and/v_or.  The test is a necessary condition of "corner at t0" evaluated at 7-bit precision (derivation at phase A); it lets
//         the same ~1/3 of the pixels through as the exact pair test.  Survivors go to an UNORDERED list by ballot compaction
//         (no prefix scan, no per-lane counters).
//   B     exact cornerScore of the listed pixels, two per lane in packed 16-bit lanes (3-input packed min / max), whenever the list is
//         full and at the end; pixels with a positive score are compacted, in place, to the corner list at the front of the list.
//   C     strict 3x3 NMS of the corners on the score tile; survivors set a bit in a row-major bitmap
//   D     output in bitmap (= row-major = reference) order: popcount, wave prefix sum, bit extraction.
// Cells with more corners than the list holds (noise) give the corner list up: NMS and compaction then scan the score tile.
#include "orbx_types.h"
#include "orbx_block.h"
#include "orbx_kernels.h"
#include "orbx_simd.h"
#include "blur_body.h"

namespace orbx {

// FAST-9/16 ring offsets (dx,dy), k = 0..15, as in OpenCV: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)
// (0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3).
// OpenCV's cornerScore (largest threshold for which the pixel is still a corner) when the pixel is a corner at threshold t0, else 0:
// max over the 16 nine-arcs of min |v - ring|, minus 1, one-sided:
// x[k] = ring value of a dark candidate, 255 - ring value of a bright one (+ kPixBias), vp = the centre treated the same way:
//   dark   max over 9-arcs of min (v - r)  =  v - min over arcs of max r
//   bright max over 9-arcs of min (r - v)  = (255 - v) - min over arcs of max (255 - r)
// so both polarities are "centre minus the smallest 9-arc maximum": 9-arc maxima as max3 of three 3-arc maxima, the minimum over the 16
// arcs by min3 (40 packed 3-input ops for two pixels).
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

// number of set bits of a wave ballot below this lane (v_mbcnt_lo/hi)
__device__ __forceinline__ int lanes_below(unsigned long long bal) {
#ifdef ORBX_EMU
    return __popcll(bal & ((1ull << lane_id()) - 1ull));
#else
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
#endif
}

// ballot of "byte J of m is not zero" (the survivor masks of phase A hold 0x00 or 0x80 per byte): one SDWA compare that selects the byte, where
// the compiler spends a v_bfe_u32 and a v_cmp per byte
template <int J>
__device__ __forceinline__ unsigned long long byte_ballot(uint32_t m) {
#ifdef ORBX_EMU
    return ORBX_BALLOT((m & (0xFFu << (8 * J))) != 0u);
#else
    unsigned long long r;
    const uint32_t zero = 0u;
    if (J == 0) asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_0 src1_sel:DWORD" : "=s"(r) : "v"(m), "v"(zero));
    else if (J == 1) asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_1 src1_sel:DWORD" : "=s"(r) : "v"(m), "v"(zero));
    else if (J == 2) asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_2 src1_sel:DWORD" : "=s"(r) : "v"(m), "v"(zero));
    else asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_3 src1_sel:DWORD" : "=s"(r) : "v"(m), "v"(zero));
    return r;
#endif
}

constexpr int kFastThreads = 64;                    // one wave per cell
static_assert(kFastThreads == kFastThreadsDecl, "launch configuration");
constexpr int kEntOff = 0x3FFF, kEntBright = 0x8000;   // survivor-list entry: tile byte offset of the pixel | bright-candidate flag
constexpr uint32_t kH = 0x80808080u, kL7 = 0x7F7F7F7Fu;

// WPC: compile-time LDS pitch of the window tile (kFastPitch for the common cells: every LDS offset of the ring becomes an immediate),
// 0 = the pitch is the cell's own window width (large cells of small pyramid levels).
template <int WPC>
__device__ __forceinline__ void fast_cell(const CellInfo ci, const LevelInfo L, const uint8_t* __restrict__ img, int iniTh, int minTh,
                                          uint32_t* __restrict__ out, int* __restrict__ count_out, uint8_t* smem, int tile_bytes, int list_bytes) {
    const int lane = (int)threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int base = 0;
    // The reference runs FAST at iniTh and, for a cell that yields nothing, again at minTh (:1135-1148).  A corner at threshold t is a pixel
    // with cornerScore >= t, and a strict 3x3 maximum with score >= t has no neighbour that a lower threshold could add above it, so the
    // first run only needs the pixels that are corners at iniTh: the quick test and the list work at t0 = iniTh (the headline workload:
    // -23 % survivors, -29 % corners against t0 = minTh; every cell has corners at 20), and only a cell that comes out empty is done again
    // at minTh, window reload included (the bitmaps of phase C have overwritten the tile by then).  Wave-uniform loop.
    int cx0 = ci.x0, cx1 = ci.x1, cy0 = ci.y0, cy1 = ci.y1;
    for (int pass = 0;; pass++) {
#ifndef ORBX_EMU
    // the cell rectangle is made opaque per run: everything derived from it is then computed inside the run, as in a straight-line kernel,
    // instead of being hoisted out of this loop and kept alive (and spilled) across all phases
    asm volatile("" : "+s"(cx0), "+s"(cx1), "+s"(cy0), "+s"(cy1));
#endif
    const int iw = cx1 - cx0, ih = cy1 - cy0, wh = ih + 6, npix = iw * ih;
    const int gx0 = (cx0 - 3) & ~3, gx1 = (cx1 + 3 + 3) & ~3;          // dword-aligned window
    const int wpr = gx1 - gx0;                              // window bytes per row
    const int wp = WPC ? WPC : wpr, wpd = wp >> 2;          // LDS pitch
    const int xo = (cx0 - 3) - gx0;                         // tile column of the window's first pixel (0..3)
    // LDS: 16 bytes of padding (the dword left of a row start is read, never used) | window tile, raw bytes | score tile | survivor / corner
    // list.  The bitmaps of phases C / D and the keep flags of the list-free NMS live in the window tile, which is dead by then.
    uint8_t* tile = smem + 16;
    uint8_t* sc = tile + tile_bytes;
    const int sc_bytes = wp * (ih + 2);
    uint16_t* list = (uint16_t*)(sc + ((sc_bytes + 15) & ~15));
#ifdef ORBX_FAST_LIST_CAP                                   // tests rebuild with a tiny capacity to force the flush / overflow paths
    const int list_cap = ORBX_FAST_LIST_CAP;
#else
    const int list_cap = list_bytes >> 1;                   // entries
#endif
    uint32_t* tile32 = (uint32_t*)tile;
    const int t0 = pass == 0 ? iniTh : minTh;
    // ---- load ----  aligned dwords of the window -> tile
    if (WPC) {
        // (WPC / 4) dword columns x 5 rows per pass, four passes (20 rows) in flight per lane
        constexpr int kCols = WPC ? WPC / 4 : 1;
        const int r0 = lane / kCols, c = lane - r0 * kCols;
        if (r0 < 5 && 4 * c < wpr) {
            const uint8_t* src = img + (uint32_t)(mul24(cy0 - 3 + r0, L.pitch) + gx0 + 4 * c);
            const uint32_t step = (uint32_t)L.pitch * 5u;
            int lo = r0 * kCols + c;
            for (int r = r0; r < wh; r += 20) {
                uint32_t v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) if (r + 5 * k < wh) v[k] = *(const uint32_t*)(src + k * step);
#pragma unroll
                for (int k = 0; k < 4; k++) if (r + 5 * k < wh) tile32[lo + 5 * k * kCols] = v[k];
                src += 4 * step; lo += 20 * kCols;
            }
        }
    } else {
        const unsigned Mw = (1u << 20) / (unsigned)wpd + 1u;    // i / wpd == (i * Mw) >> 20 exactly for i < 2^13
        for (int i0 = lane; i0 < wh * wpd; i0 += 4 * kFastThreads) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = i0 + k * kFastThreads;
                if (i < wh * wpd) {
                    const int r = (int)((unsigned)mul24(i, (int)Mw) >> 20), c = i - mul24(r, wpd);
                    v[k] = *(const uint32_t*)(img + (uint32_t)(mul24(cy0 - 3 + r, L.pitch) + gx0 + 4 * c));
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) { const int i = i0 + k * kFastThreads; if (i < wh * wpd) tile32[i] = v[k]; }
        }
    }
    // score tile: same pitch as the window tile, rows -1 .. ih of the interior (one zero row / column around it), so that
    // score offset = tile offset - 2 * wp and the 3x3 NMS reads its 8 neighbours at fixed offsets without bounds tests
    for (int i = lane; i < (sc_bytes + 3) >> 2; i += kFastThreads) ((uint32_t*)sc)[i] = 0u;
    ORBX_WAVE_SYNC();
    // exact score of two (pixel, polarity) entries in the two halves of packed registers
    auto score2 = [&](int eA, int eB, int& sA, int& sB) {
        // a bright candidate is scored on the inverted image (byte ^ 0xFF); the bias that makes every operand a positive binary16 pattern
        // does not overlap the pixel bits, so one xor applies both
        const uint32_t X = (uint32_t)kPixBias * 0x00010001u | ((eA & kEntBright) ? 0x000000FFu : 0u) | ((eB & kEntBright) ? 0x00FF0000u : 0u);
        // window pixel (-3, -3) of the candidate: every ring offset from it is >= 0 and fits the ds_read offset field (the offsets are made
        // opaque so that the compiler keeps this one base per candidate instead of re-deriving bases for the rows above the centre)
        int ofa = (eA & kEntOff) - 3 * wp - 3, ofb = (eB & kEntOff) - 3 * wp - 3;
#ifndef ORBX_EMU
        asm volatile("" : "+v"(ofa), "+v"(ofb));
#endif
        const uint8_t* qa = tile + ofa;
        const uint8_t* qb = tile + ofb;
        pk2 d[16];
#define ORBX_D(k, dx, dy) d[k] = pk_xor(pk_bytes(qa + ((dy) + 3) * wp + (dx) + 3, qb + ((dy) + 3) * wp + (dx) + 3), X);
        ORBX_D(0, 0, 3)     ORBX_D(1, 1, 3)     ORBX_D(2, 2, 2)     ORBX_D(3, 3, 1)
        ORBX_D(4, 3, 0)     ORBX_D(5, 3, -1)    ORBX_D(6, 2, -2)    ORBX_D(7, 1, -3)
        ORBX_D(8, 0, -3)    ORBX_D(9, -1, -3)   ORBX_D(10, -2, -2)  ORBX_D(11, -3, -1)
        ORBX_D(12, -3, 0)   ORBX_D(13, -3, 1)   ORBX_D(14, -2, 2)   ORBX_D(15, -1, 3)
        const pk2 vp = pk_xor(pk_bytes(qa + 3 * wp + 3, qb + 3 * wp + 3), X);
#undef ORBX_D
        fast_score_pk(d, vp, t0, sA, sB);
    };
    // The list region holds [0, nc) corner entries (score offsets of pixels with a positive score) followed by the pending survivors
    // [pbeg, cnt) of phase A.  ---- B ----  scores the pending entries, two per lane, and compacts the corners among them in place: a trip
    // reads 128 entries and appends at most that many behind entries that have been consumed.  When the corners alone no longer leave room
    // for a phase-A trip, the corner list is given up (corners_listed = false) and phases C / D scan the score tile instead.
    int nc = 0, pbeg = 0, cnt = 0;
    bool corners_listed = true;
    auto score_pending = [&]() {
        for (int i0 = pbeg; i0 < cnt; i0 += 2 * kFastThreads) {
            const int i = i0 + 2 * lane;
            int sA = 0, sB = 0, oA = 0, oB = 0;
            const bool hasB = i + 1 < cnt;
            if (i < cnt) {
                const uint32_t ee = *(const uint32_t*)(list + i);       // pbeg is even
                const int eA = (int)(ee & 0xFFFFu), eB = hasB ? (int)(ee >> 16) : eA;
                score2(eA, eB, sA, sB);
                oA = (eA & kEntOff) - 2 * wp; oB = (eB & kEntOff) - 2 * wp;
                // a pixel listed for both polarities can be a corner for at most one of them (two 9-arcs of opposite sign do not fit on 16
                // ring pixels), so only a positive score is written
                if (sA > 0) sc[oA] = (uint8_t)sA;
                if (hasB && sB > 0) sc[oB] = (uint8_t)sB;
            }
            if (corners_listed) {
                const bool cA = sA > 0, cB = hasB && sB > 0;
                const unsigned long long balA = ORBX_BALLOT(cA), balB = ORBX_BALLOT(cB);
                const int nA = __popcll(balA);
                if (ORBX_IN_BALLOT(balA)) list[nc + lanes_below(balA)] = (uint16_t)oA;
                if (ORBX_IN_BALLOT(balB)) list[nc + nA + lanes_below(balB)] = (uint16_t)oB;
                nc += nA + __popcll(balB);
            }
        }
        ORBX_WAVE_SYNC();
        pbeg = cnt = corners_listed ? (nc + 1) & ~1 : 0;
    };
    // ---- A ----  quick rejection, 4 adjacent pixels per lane, 7-bit SWAR on the "H form" (p >> 1) | 0x80 of every byte.
    // A dark 9-arc holds one point of every opposite ring pair, so a dark corner at t0 needs, for each of the four pairs (0,8) (2,10) (4,12)
    // (6,14), a ring pixel with r < v - t0, i.e. r <= v - t0 - 1; a bright one needs r >= v + t0 + 1.  With R = r >> 1, V = v >> 1,
    // td = (t0 + 1) >> 1:   r <= v - (t0 + 1)  =>  R <= floor((v - (t0 + 1)) / 2) <= V - td     ( floor(a - b) <= floor(a) - floor(b) )
    //                       r >= v + (t0 + 1)  =>  R >= floor((v + (t0 + 1)) / 2) >= V + td .
    // Per byte, with the guard bit of the H form:  (R | 0x80) - Q  has its top bit set  <=>  R >= Q  (0 <= Q <= 128, no borrow leaves the
    // byte).  Q = max(V - td + 1, 0) gives "not dark", Q = min(V + td, 128) gives "bright"; the saturations cost a few more SWAR ops on the
    // centre dword.  The test passes everything the exact pair test passes, plus v - r = t0 (+1) for some parities (< 1 % more pixels).
    // (x >> 1) | 0x80808080 is the H form of all four bytes: the bit that leaks in from the neighbouring byte lands on the guard bit.
    {
        const int g0 = (xo + 3) >> 2, ng = ((xo + 3 + iw - 1) >> 2) - g0 + 1;       // dword groups of a row (<= 61: cells are < 244 px wide)
        // A trip covers rpt = 64 / ng whole rows: lane = (row within the trip) * ng + (dword group), so a lane keeps its group - and with it
        // its x position, the validity of its four pixels and its tile column - for the whole cell, and only the row advances.
        const int rpt = 64 / ng;
        const int yl = lane / ng, gi = lane - yl * ng;
        const bool lane_used = yl < rpt;
        const int g = g0 + gi;
        const int xbase = 4 * g - (xo + 3);                     // interior x of this lane's first pixel (may be < 0)
        uint32_t VM = 0;                                        // top bit of byte j: pixel j of this lane is an interior pixel
#pragma unroll
        for (int j = 0; j < 4; j++) if (lane_used && xbase + j >= 0 && xbase + j < iw) VM |= 0x80u << (8 * j);
        const int td = (t0 + 1) >> 1;
        const uint32_t Cd = (uint32_t)(td - 1) * 0x01010101u, Tb = (uint32_t)td * 0x01010101u;
        const uint32_t* hb_lane = tile32 + yl * wpd + (g - 1);  // window row (interior row - 3), dword g - 1
        const int toff_lane = (yl + 3) * wp + 4 * g;            // tile byte offset of the lane's first pixel in trip 0
        for (int it0 = 0; it0 < ih; it0 += rpt) {
            uint32_t SD = 0, SB = 0;
            if (VM != 0 && it0 + yl < ih) {
                if (t0 >= 1) {
                    const uint32_t* hb = hb_lane + it0 * wpd;
                    const uint32_t C0 = hb[1], L1 = hb[wpd], C1 = hb[wpd + 1], R1 = hb[wpd + 2], L3 = hb[3 * wpd], C3 = hb[3 * wpd + 1], R3 = hb[3 * wpd + 2],
                                   L5 = hb[5 * wpd], C5 = hb[5 * wpd + 1], R5 = hb[5 * wpd + 2], C6 = hb[6 * wpd + 1];
#define ORBX_H(w) (((w) >> 1) | kH)
                    // thresholds from the centre dword
                    const uint32_t c3 = ORBX_H(C3);
                    const uint32_t x = c3 - Cd;                     // 128 + V - (td - 1) >= 1
                    const uint32_t mx = x & kH;
                    const uint32_t Qd = x & (mx - (mx >> 7));       // max(V - td + 1, 0)
                    const uint32_t y = (c3 & kL7) + Tb;             // V + td <= 255
                    const uint32_t my = y & kH;
                    const uint32_t Pb = y ^ (y & (my - (my >> 7))); // min(V + td, 128)
                    // ring pixels k = 0, 2, .., 14 of the four pixels (index k / 2)
                    const uint32_t q0 = ORBX_H(C6), q1 = ORBX_H(align_byte(R5, C5, 2)), q2 = ORBX_H(align_byte(R3, C3, 3)), q3 = ORBX_H(align_byte(R1, C1, 2)),
                                   q4 = ORBX_H(C0), q5 = ORBX_H(align_byte(C1, L1, 2)), q6 = ORBX_H(align_byte(C3, L3, 1)), q7 = ORBX_H(align_byte(C5, L5, 2));
#undef ORBX_H
                    // top bit of (q - Qd): ring pixel is NOT dark enough; of (q - Pb): ring pixel is bright enough
                    const uint32_t nd = ((q0 - Qd) & (q4 - Qd)) | ((q1 - Qd) & (q5 - Qd)) | ((q2 - Qd) & (q6 - Qd)) | ((q3 - Qd) & (q7 - Qd));
                    const uint32_t sb = ((q0 - Pb) | (q4 - Pb)) & ((q1 - Pb) | (q5 - Pb)) & ((q2 - Pb) | (q6 - Pb)) & ((q3 - Pb) | (q7 - Pb));
                    SD = (nd & VM) ^ VM;
                    SB = sb & VM;
                } else SD = SB = VM;                                // threshold 0: the byte arithmetic needs td >= 1; every pixel is scored
            }
            const uint32_t ANY = SD | SB, BOTH = SD & SB;
            unsigned long long bal[4], bald[4];
            int trip = 0;
            // (the appends below are guarded by ORBX_IN_BALLOT(bal[j]), not by a lane predicate: the compiler tested every bit twice otherwise, as
            // and + cmp for the branch and as bfe + cmp for the ballot)
            bal[0] = byte_ballot<0>(ANY); bal[1] = byte_ballot<1>(ANY); bal[2] = byte_ballot<2>(ANY); bal[3] = byte_ballot<3>(ANY);
#pragma unroll
            for (int j = 0; j < 4; j++) trip += __popcll(bal[j]);
            const bool dups = ORBX_BALLOT(BOTH != 0u) != 0ull;      // a pixel that passes for both polarities gets a second (dark) entry: rare
            if (dups) {
                bald[0] = byte_ballot<0>(BOTH); bald[1] = byte_ballot<1>(BOTH); bald[2] = byte_ballot<2>(BOTH); bald[3] = byte_ballot<3>(BOTH);
#pragma unroll
                for (int j = 0; j < 4; j++) trip += __popcll(bald[j]);
            }
            if (cnt + trip > list_cap) {                         // wave-uniform: score what is pending, then append behind the corners
                score_pending();
                if (cnt + trip > list_cap) { corners_listed = false; nc = 0; pbeg = cnt = 0; }      // (a trip adds <= 512 <= list_cap entries)
            }
            const int toff = toff_lane + it0 * wp;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (ORBX_IN_BALLOT(bal[j])) list[cnt + lanes_below(bal[j])] = (uint16_t)((toff + j) | ((SB & (0x80u << (8 * j))) ? kEntBright : 0));
                cnt += __popcll(bal[j]);
            }
            if (dups) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (ORBX_IN_BALLOT(bald[j])) list[cnt + lanes_below(bald[j])] = (uint16_t)(toff + j);
                    cnt += __popcll(bald[j]);
                }
            }
        }
    }
    ORBX_WAVE_SYNC();
    score_pending();
    if (corners_listed) {
        // ---- C ----  cell-local strict 3x3 NMS of the corners (every listed pixel has a score >= t0); survivors mark their (row-major)
        // score offset in a bitmap
        uint32_t* bm = tile32;                                         // bit o: corner at score offset o survived
        const int nwords = (sc_bytes + 31) >> 5;
        for (int i = lane; i < nwords; i += kFastThreads) bm[i] = 0u;
        ORBX_WAVE_SYNC();
        for (int i = lane; i < nc; i += kFastThreads) {
            const int o = list[i];
            const uint8_t* c = sc + o;
            const int s = c[0];
            // neighbours outside the cell interior are the zero frame
            const int m0 = imax(imax((int)c[-wp - 1], (int)c[-wp]), imax((int)c[-wp + 1], (int)c[-1]));
            const int m1 = imax(imax((int)c[1], (int)c[wp - 1]), imax((int)c[wp], (int)c[wp + 1]));
            if (s > imax(m0, m1)) atomicOr(&bm[o >> 5], 1u << (o & 31));
        }
        ORBX_WAVE_SYNC();
        // ---- D ----  output in bitmap order = row-major order of the pixels
        for (int w0 = 0; w0 < nwords; w0 += kFastThreads) {
            const int w = w0 + lane;
            uint32_t bits = w < nwords ? bm[w] : 0u;
            const int c = __popc(bits);
            const int incl = wave_incl_scan(c);
            int pos = base + incl - c;
            while (bits) {
                const int b = __ffsll((unsigned long long)bits) - 1;
                bits &= bits - 1u;
                const int o = 32 * w + b;
                const int y1 = o / (WPC ? WPC : wp), col = o - y1 * wp;      // score row (interior y + 1), tile column
                out[pos++] = key_pack(cx0 + (col - (xo + 3)) - kBorder, cy0 + (y1 - 1) - kBorder, (int)sc[o]);
            }
            base += ORBX_READLANE(incl, 63);
        }
    } else {
        // ---- C', D' ----  the same over every interior pixel (cells so dense that their corners do not fit the list)
        const unsigned Mi = (1u << 20) / (unsigned)iw + 1u;     // i / iw == (i * Mi) >> 20 exactly for i < 2^13
        uint8_t* kf = tile;                                     // one keep flag per score-tile byte
        for (int i = lane; i < npix; i += kFastThreads) {
            const int y = (int)((unsigned)mul24(i, (int)Mi) >> 20), x = i - mul24(y, iw);
            const int o = (y + 1) * wp + xo + 3 + x;
            const uint8_t* c = sc + o;
            const int s = c[0];
            const int m0 = imax(imax((int)c[-wp - 1], (int)c[-wp]), imax((int)c[-wp + 1], (int)c[-1]));
            const int m1 = imax(imax((int)c[1], (int)c[wp - 1]), imax((int)c[wp], (int)c[wp + 1]));
            kf[o] = (uint8_t)(s > imax(m0, m1));
        }
        ORBX_WAVE_SYNC();
        for (int i0 = 0; i0 < npix; i0 += kFastThreads) {
            const int i = i0 + lane;
            int flag = 0, x = 0, y = 0, s = 0;
            if (i < npix) {
                y = (int)((unsigned)mul24(i, (int)Mi) >> 20); x = i - mul24(y, iw);
                const int o = (y + 1) * wp + xo + 3 + x;
                s = sc[o];
                flag = kf[o];
            }
            const unsigned long long bal = ORBX_BALLOT(flag);
            if (flag) out[base + __popcll(bal & lt)] = key_pack(cx0 + x - kBorder, cy0 + y - kBorder, s);
            base += __popcll(bal);
        }
    }
    if (base > 0 || pass == 1 || minTh >= iniTh) break;
    ORBX_WAVE_SYNC();                                           // the second run reloads the window over the bitmap / keep flags
    }
    if (lane == 0) *count_out = base;
}

// One single-wave workgroup per (cell, image).  slots: per-cell candidate lists in the reference order; cell_count[b * ncells + cell] =
// number kept.  dynamic LDS = 16 + tile_bytes + score-tile bytes + list_bytes.  bx = the workgroup's index among the FAST workgroups of its image.
__device__ __forceinline__ void fast_block(int bx, const LevelInfo* __restrict__ lv, const CellInfo* __restrict__ cells, int ncells,
                                           const uint8_t* __restrict__ pyr, size_t pyr_stride, int iniTh, int minTh,
                                           uint32_t* __restrict__ slots, size_t slots_stride, int* __restrict__ cell_count, int tile_bytes, int list_bytes,
                                           int* __restrict__ status, uint8_t* smem) {
    // the batch's status word (the quadtree's capacity flag) is cleared here, by the kernel in front of the quadtree, instead of by a fill launch
    if (bx == 0 && blockIdx.y == 0 && threadIdx.x < 4) status[threadIdx.x] = 0;
    // Workgroup -> cell mapping.  Consecutive workgroup ids go to different XCDs (id % 8), each with its own L2; with the plain mapping
    // (workgroup b -> cell b) neighbouring cells never share an L2 and the 6-pixel window overlap plus the dword / cache-line padding of
    // every window row is fetched again per cell.  Runs of kFastXcdRun neighbouring cells are therefore kept on one XCD
    // (run length 1 / 2 / 4 / 20: 305 / 186 / 131 / 80 MB fetched per 128 images; long runs skew the mix of dense and sparse cells per XCD).
    const int xcd = bx & 7, jj = bx >> 3;
    const int cell = ((jj / kFastXcdRun) * 8 + xcd) * kFastXcdRun + (jj % kFastXcdRun), b = (int)blockIdx.y;
    if (cell >= ncells) return;
    const CellInfo ci = cells[cell];
    const LevelInfo L = lv[ci.level];
    int* count_out = cell_count + (size_t)b * ncells + cell;
    if (ci.x1 - ci.x0 <= 0 || ci.y1 - ci.y0 <= 0) {
        if (threadIdx.x == 0) *count_out = 0;
        return;
    }
    const uint8_t* img = pyr + (size_t)b * pyr_stride + L.off;
    uint32_t* out = slots + (size_t)b * slots_stride + ci.slot_off;
    const int wpr = ((ci.x1 + 3 + 3) & ~3) - ((ci.x0 - 3) & ~3);
    if (wpr <= kFastPitch) fast_cell<kFastPitch>(ci, L, img, iniTh, minTh, out, count_out, smem, tile_bytes, list_bytes);
    else fast_cell<0>(ci, L, img, iniTh, minTh, out, count_out, smem, tile_bytes, list_bytes);
}

__global__ void __launch_bounds__(kFastThreads) k_fast_cells(const LevelInfo* __restrict__ lv,
                                                    const CellInfo* __restrict__ cells, int ncells,
                                                    const uint8_t* __restrict__ pyr, size_t pyr_stride,
                                                    int iniTh, int minTh,
                                                    uint32_t* __restrict__ slots, size_t slots_stride,
                                                    int* __restrict__ cell_count, int tile_bytes, int list_bytes, int* __restrict__ status) {
    ORBX_DYN_SMEM(smem);
    fast_block((int)blockIdx.x, lv, cells, ncells, pyr, pyr_stride, iniTh, minTh, slots, slots_stride, cell_count, tile_bytes, list_bytes, status, smem);
}

// Small batches (one pair per call: Tracking's rhythm): the blur of the pyramid depends on the pyramid alone, like FAST, and at large batches
// runs beside it on a second stream.  A fork and a join of two streams cost a kernel behind each of them ~6 us (barrier packets), 12 of the 166 us
// of a stereo pair - so here the two share ONE launch instead: the first `blur_waves` single-wave workgroups of an image blur one strip each
// (blur_body.h; no LDS, the launch's allocation goes unused), the others are the FAST cells.  Both halves are the kernels above, unchanged.
__global__ void __launch_bounds__(kFastThreads) k_fast_cells_blur(const LevelInfo* __restrict__ lv,
                                                    const CellInfo* __restrict__ cells, int ncells,
                                                    const uint8_t* __restrict__ pyr, size_t pyr_stride,
                                                    int iniTh, int minTh,
                                                    uint32_t* __restrict__ slots, size_t slots_stride,
                                                    int* __restrict__ cell_count, int tile_bytes, int list_bytes, int* __restrict__ status,
                                                    int nlevels, uint8_t* __restrict__ blur, BlurTaps taps, BlurTiles tiles, int blur_waves) {
    ORBX_DYN_SMEM(smem);
    const int bx = (int)blockIdx.x;
    if (bx < blur_waves) { blur_strip<kBlurRows>(lv, nlevels, pyr, blur, pyr_stride, taps, tiles, bx >> 2, bx & 3, (int)threadIdx.x, (int)blockIdx.y); return; }
    fast_block(bx - blur_waves, lv, cells, ncells, pyr, pyr_stride, iniTh, minTh, slots, slots_stride, cell_count, tile_bytes, list_bytes, status, smem);
}

}  // namespace orbx
