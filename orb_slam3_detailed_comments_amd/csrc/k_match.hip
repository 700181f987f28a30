// k_match.hip — descriptor matching kernels (integer/bitwise; wave64 popcount + min reductions):
//   k_hamming_matrix  all-pairs ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2383-2403)
//   k_stereo_match    Frame::ComputeStereoMatches steps 1-4 (src/Frame.cc:1102-1340): row-band
//                     candidate search, best Hamming, 11x11 SAD over 11 shifts, parabola sub-pixel fit
//   k_stereo_median   step 5 (:1343-1357): reject matches with SAD >= 1.5*1.4*median
//   k_knn2            BFMatcher(NORM_HAMMING).knnMatch(k=2) + Lowe ratio of
//                     Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1553-1562)
// Descriptors are read as 4 x u64 (the reference reads 8 x int32; the popcount sum is identical).
#include "orbx_types.h"
#include "orbx_block.h"
#include "orbx_kernels.h"
#include "orbx_simd.h"
#include "kb8_model.h"

namespace orbx {

__device__ __forceinline__ int hamming256(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b) {
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

// out[i*nb + j] = distance(descA[i], descB[j]);  grid (ceil(nb/256), na)
__global__ void __launch_bounds__(256) k_hamming_matrix(const unsigned long long* __restrict__ A, int na,
                                                        const unsigned long long* __restrict__ Bm, int nb,
                                                        int* __restrict__ out) {
    const int i = (int)blockIdx.y, j = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= na || j >= nb) return;
    out[(size_t)i * nb + j] = hamming256(A + 4 * (size_t)i, Bm + 4 * (size_t)j);
}

// grid (ceil(cap/4), B); one wave per left keypoint.
// kpsL/descL/nL: left extractor outputs (stride cap); same for right; pyrL/pyrR: raw pyramids; bucket_start / bucket_items: the row index
// of the right keypoints (k_layout).
__global__ void __launch_bounds__(256) k_stereo_match(const LevelInfo* __restrict__ lv,
                                                      const KeyPointRec* __restrict__ kpsL, const unsigned long long* __restrict__ descL, const int* __restrict__ nL,
                                                      const KeyPointRec* __restrict__ kpsR, const unsigned long long* __restrict__ descR,
                                                      const int4* __restrict__ auxR, const int* __restrict__ nR,
                                                      const int* __restrict__ bucket_start, const int* __restrict__ bucket_items, int nb, int lookback,
                                                      int cap, const uint8_t* __restrict__ pyrL, const uint8_t* __restrict__ pyrR, size_t pyr_stride,
                                                      StereoParams P, float* __restrict__ uRight, float* __restrict__ depth, int* __restrict__ sad) {
    const int b = (int)blockIdx.y, lane = lane_id();
    const int iL = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (iL >= cap) return;
    const size_t o = (size_t)b * cap + iL;
    if (iL >= nL[b]) { if (lane == 0) { uRight[o] = -1.0f; depth[o] = -1.0f; sad[o] = -1; } return; }
    const KeyPointRec kL = kpsL[o];
    const int levelL = kL.octave;
    const float uL = kL.x, vL = kL.y;
    const float maxD = __fdiv_rn(P.mbf, P.mb);
    const float minU = __fsub_rn(uL, maxD), maxU = uL;
    float out_u = -1.0f, out_d = -1.0f; int out_sad = -1;
    const int rowL = __float2int_rz(vL);
    const int nr = nR[b];
    // key = distance << 16 | iR.  The reference starts from bestDist = TH_HIGH, bestIdxR = 0 and replaces on a strictly smaller distance
    // while it walks vRowIndices[vL] in ascending iR (src/Frame.cc:1195-1226): the winner is the smallest key below TH_HIGH << 16.
    unsigned best = (unsigned)P.th_high << 16;
    if (!(maxU < 0)) {
        const unsigned long long* dl = descL + 4 * o;
        const unsigned long long d0 = dl[0], d1 = dl[1], d2 = dl[2], d3 = dl[3];
        const int4* ar = auxR + (size_t)b * cap;          // {first row, last row, x bits, octave}: one coalesced 16-B load per candidate
        // candidates: right keypoints whose band starts in rows [rowL - lookback, rowL] (lookback >= the tallest band), i.e. a
        // contiguous run of row buckets; the exact band / octave / column tests follow
        const int* bs = bucket_start + (size_t)b * (nb + 1);
        const int* items = bucket_items + (size_t)b * cap;
        const int jbeg = bs[imin(imax(rowL - lookback, 0) >> kStereoRowShift, nb - 1)], jend = bs[imin(imax(rowL, 0) >> kStereoRowShift, nb - 1) + 1];
        // one candidate: exact band / octave / column tests, then the (distance, index) key
        auto visit = [&](int iR, const int4& a) {
            if (rowL < a.x || rowL > a.y) return;
            if (a.w < levelL - 1 || a.w > levelL + 1) return;
            const float xr = __int_as_float(a.z);
            if (xr >= minU && xr <= maxU) {
                const unsigned long long* dr = descR + 4 * ((size_t)b * cap + iR);
                const int dist = __popcll(d0 ^ dr[0]) + __popcll(d1 ^ dr[1]) + __popcll(d2 ^ dr[2]) + __popcll(d3 ^ dr[3]);
                const unsigned cand = ((unsigned)dist << 16) | (unsigned)iR;
                // full-key compare: lowest distance, then lowest iR, whatever the visiting order.  (Test switch bit 1 = the
                // distance-only compare this kernel had in round 1, kept so that the tie tests can show they would catch it.)
                if ((P.debug_flags & 2) ? dist < (int)(best >> 16) : cand < best) best = cand;
            }
        };
        if (P.debug_flags & 4) {
            // test switch: every lane walks ALL candidates, last to first - every tie then meets in one lane, the higher position first
            for (int j = jend - 1; j >= jbeg; j--) { const int iR = items[j]; visit(iR, ar[iR]); }
        } else
        for (int base = jbeg; base < jend; base += 128) { // 2 right keypoints per lane per trip, their loads in flight together
            int4 a[2]; int idx[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                int j = base + 64 * u + lane;
                const bool in = j < jend;
                if (P.debug_flags & 1) j = jend - 1 - (j - jbeg);    // test switch: visit the candidates in the opposite order
                idx[u] = in ? items[j] : -1;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (idx[u] >= 0) a[u] = ar[idx[u]]; else { a[u].x = 1; a[u].y = 0; a[u].z = 0; a[u].w = 0; }   // empty band
            }
#pragma unroll
            for (int u = 0; u < 2; u++) visit(idx[u], a[u]);
        }
    }
    best = wave_min_u32(best);   // lowest distance, then lowest right index == sequential first-min
    const int bestDist = (int)(best >> 16);
    if (bestDist < P.th_orb && nr > 0) {
        const int bestIdxR = (int)(best & 0xFFFFu);
        const float uR0 = kpsR[(size_t)b * cap + bestIdxR].x;
        const LevelInfo Lv = lv[levelL];
        const float sf = Lv.inv_scale;
        const float scaleduL = roundf(__fmul_rn(kL.x, sf)), scaledvL = roundf(__fmul_rn(kL.y, sf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const int w = 5, Lh = 5;
        const float iniu = scaleduR0 + (float)(Lh - w), endu = scaleduR0 + (float)(Lh + w + 1);
        const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
        // the reference only guards iniu/endu (:1265); the extra clause keeps every read inside the level
        const bool inb = cv - w >= 0 && cv + w < Lv.h && cu - w >= 0 && cu + w < Lv.w && cr - Lh - w >= 0 && cr + Lh + w < Lv.w;
        if (!(iniu < 0 || endu >= (float)Lv.w) && inb) {
            const uint8_t* IL = pyrL + (size_t)b * pyr_stride + Lv.off;
            const uint8_t* IR = pyrR + (size_t)b * pyr_stride + Lv.off;
            // SAD of the 11 x 11 left window against the right window at the 11 shifts (:1270-1290).  Lane = (row group g = lane >> 4, shift
            // k = lane & 15 < 11) sums rows g, g + 4, g + 8 for its shift: a row is 11 bytes = three dwords at columns 0, 4 and 7 (the third
            // overlaps the second by one byte, which is masked on both sides: no byte outside the windows is read), three v_sad_u8 per row.
            // The four row groups are added across the lanes k, k + 16, k + 32, k + 48; the first minimum over k (:1283 keeps a strictly
            // smaller SAD while k ascends) is the smallest key sad << 4 | k.
            const int g = lane >> 4, k = lane & 15;
            uint32_t acc = 0;
            if (k < 11) {
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int row = g + 4 * t;
                    if (row < 11) {
                        const int rofs = __mul24(cv - w + row, Lv.pitch);                             // rows and pitches are far below 2^23
                        const uint8_t* lp = IL + (uint32_t)(rofs + (cu - w));
                        const uint8_t* rp = IR + (uint32_t)(rofs + (cr - w - Lh + k));
                        const uint32_t l0 = load_u32_any(lp), l1 = load_u32_any(lp + 4), l2 = load_u32_any(lp + 7) & 0xFFFFFF00u;
                        const uint32_t r0 = load_u32_any(rp), r1 = load_u32_any(rp + 4), r2 = load_u32_any(rp + 7) & 0xFFFFFF00u;
                        acc = sad4_u8(l0, r0, sad4_u8(l1, r1, sad4_u8(l2, r2, acc)));
                    }
                }
            }
            acc += __shfl_xor(acc, 16);
            acc += __shfl_xor(acc, 32);
            const unsigned skey = wave_min_u32(k < 11 ? (acc << 4) | (unsigned)k : 0xFFFFFFFFu);
            const int bestS = (int)(skey >> 4), bi = (int)(skey & 15u), bestinc = bi - Lh;
            if (!(bestinc == -Lh || bestinc == Lh)) {
                const float dist1 = (float)ORBX_READLANE(acc, bi - 1), dist2 = (float)bestS, dist3 = (float)ORBX_READLANE(acc, bi + 1);
                const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3),
                                               __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2))));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = __fmul_rn(Lv.scale, __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
                    float disparity = __fsub_rn(uL, bestuR);
                    if (disparity >= 0 && disparity < maxD) {
                        if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }
                        out_d = __fdiv_rn(P.mbf, disparity);
                        out_u = bestuR;
                        out_sad = bestS;
                    }
                }
            }
        }
    }
    if (lane == 0) { uRight[o] = out_u; depth[o] = out_d; sad[o] = out_sad; }
}

// grid (B), 256 threads.  The median of the reference (:1335-1338: sort the (SAD, index) pairs, take element size / 2) is the SAD of rank
// cnt / 2; a SAD of an 11 x 11 window is below 2^15, so the rank is located by two 256-bin histograms (high byte, then low 7 bits inside the
// bin that holds the rank) instead of comparing every pair of values.
__global__ void __launch_bounds__(256) k_stereo_median(const int* __restrict__ nL, int cap,
                                                       float* __restrict__ uRight, float* __restrict__ depth,
                                                       const int* __restrict__ sad, int* __restrict__ n_matches) {
    __shared__ unsigned long long s_scan[20];
    __shared__ int s_hist[256];
    __shared__ int s_med[3];                                // {median, bin of the rank, rank inside the bin}
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
    const int n = nL[b];
    const int* sd = sad + (size_t)b * cap;
    int med = -1;
    for (int pass = 0; pass < 2; pass++) {
        s_hist[tid] = 0;
        __syncthreads();
        const int bin = pass ? s_med[1] : 0;
        for (int i = tid; i < n; i += 256) {
            const int s = sd[i];
            if (s >= 0 && (pass == 0 || (s >> 7) == bin)) atomicAdd(&s_hist[pass ? s & 127 : imin(s >> 7, 255)], 1);
        }
        __syncthreads();
        unsigned long long tot;
        const int c = s_hist[tid];
        const int ex = (int)block_excl_scan<unsigned long long>((unsigned long long)c, &tot, s_scan);
        const int cnt = pass ? 0 : (int)tot;
        if (pass == 0 && cnt == 0) { if (tid == 0) n_matches[b] = 0; return; }
        const int k = pass ? s_med[2] : cnt / 2;
        if (ex <= k && k < ex + c) {                         // exactly one bin holds the rank
            if (pass == 0) { s_med[1] = tid; s_med[2] = k - ex; }
            else s_med[0] = (bin << 7) | tid;
        }
        __syncthreads();
    }
    med = s_med[0];
    const float median = (float)med;
    const float thDist = __fmul_rn(1.5f * 1.4f, median);
    int kept = 0;
    for (int i = tid; i < n; i += 256) {
        const int s = sad[(size_t)b * cap + i];
        if (s >= 0) {
            if (!((float)s < thDist)) { uRight[(size_t)b * cap + i] = -1.0f; depth[(size_t)b * cap + i] = -1.0f; }
            else kept++;
        }
    }
    kept = wave_sum(kept);
    if ((tid & 63) == 0) s_scan[tid >> 6] = (unsigned long long)kept;
    __syncthreads();
    if (tid == 0) n_matches[b] = (int)(s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3]);
}

// grid (ceil(cap/4), B); wave per query.  Query rows [qoff[b], nq[b]) of descQ, train rows [toff[b], nt[b]) of descT.
// Outputs are indexed by query row relative to qoff; indices are relative to toff (like DMatch.trainIdx).
__global__ void __launch_bounds__(256) k_knn2(const unsigned long long* __restrict__ descQ, const int* __restrict__ qoff, const int* __restrict__ nq,
                                              const unsigned long long* __restrict__ descT, const int* __restrict__ toff, const int* __restrict__ nt,
                                              int cap, int* __restrict__ idx0, int* __restrict__ dist0, int* __restrict__ idx1, int* __restrict__ dist1,
                                              uint8_t* __restrict__ ratio_ok) {
    const int b = (int)blockIdx.y, lane = lane_id();
    const int q = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int q0 = qoff ? qoff[b] : 0, t0 = toff ? toff[b] : 0;
    const int nQ = nq[b] - q0, nT = nt[b] - t0;
    if (q >= cap) return;
    const size_t o = (size_t)b * cap + q;
    if (q >= nQ) { if (lane == 0) { idx0[o] = -1; idx1[o] = -1; dist0[o] = -1; dist1[o] = -1; ratio_ok[o] = 0; } return; }
    const unsigned long long* dq = descQ + 4 * ((size_t)b * cap + q0 + q);
    const unsigned long long a0 = dq[0], a1 = dq[1], a2 = dq[2], a3 = dq[3];
    unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;   // (dist << 16 | idx), lexicographic == strict '<' insertion order
    for (int j = lane; j < nT; j += 64) {
        const unsigned long long* dt = descT + 4 * ((size_t)b * cap + t0 + j);
        const int d = __popcll(a0 ^ dt[0]) + __popcll(a1 ^ dt[1]) + __popcll(a2 ^ dt[2]) + __popcll(a3 ^ dt[3]);
        const unsigned key = ((unsigned)d << 16) | (unsigned)j;
        if (key < k0) { k1 = k0; k0 = key; } else if (key < k1) k1 = key;
    }
    const unsigned b0 = wave_min_u32(k0);
    if (k0 == b0) k0 = k1;            // the winner's lane exposes its runner-up
    const unsigned b1 = wave_min_u32(k0);
    if (lane == 0) {
        const int i0 = b0 == 0xFFFFFFFFu ? -1 : (int)(b0 & 0xFFFF), i1 = b1 == 0xFFFFFFFFu ? -1 : (int)(b1 & 0xFFFF);
        const int dd0 = i0 < 0 ? -1 : (int)(b0 >> 16), dd1 = i1 < 0 ? -1 : (int)(b1 >> 16);
        idx0[o] = i0; idx1[o] = i1; dist0[o] = dd0; dist1[o] = dd1;
        ratio_ok[o] = (i1 >= 0 && (double)(float)dd0 < (double)(float)dd1 * 0.7) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------------
// Brute-force 2-NN on the matrix cores.  With the descriptor bits as +1 / -1 bytes, the dot product of two descriptors is 256 - 2 * Hamming
// distance: the all-pairs distance matrix of a query set against a train set is an i8 GEMM (K = 256), and the 2-NN is a row-wise top-2 over it.
//
// k_knn2_mfma: same outputs as k_knn2.  grid (ceil(cap / 32), B), 256 threads: the four waves of a workgroup share 32 queries (the B operand of
// v_mfma_i32_32x32x32_i8: 8 k-steps x 16 bytes per lane, built once per wave) and split the train rows in tiles of 32 (the A operand; wave w takes
// tiles w, w + 4, ..), then merge their top-2 through LDS.  The descriptors stay packed in memory: lane (row r = l & 31, half h = l >> 5) loads
// the 16 bytes [16 h, 16 h + 16) of its row - ONE dwordx4 per lane and tile, 1 KB per tile instead of 8 KB of expanded bytes - and expands 16 bits
// per k-step through a 16-entry nibble -> dword table in LDS (element (k-step s, half h, j) of the MFMA's K = 32 is bit 128 h + 16 s + j of the
// descriptor: any assignment works as long as both operands use it).  D[train][query] lands with the query in the lane (column l & 31) and 16
// train rows per lane in the accumulator registers, so the top-2 update is lane-local:
//   key = (256 - D) << 15 | train row = Hamming distance << 16 | train row,   k1 = min(k1, max(k0, key)),   k0 = min(k0, key)
// - the lexicographic (distance, index) order of k_knn2 (strict '<' while the index ascends).  The two lanes of a query (l, l + 32: the two halves
// of a tile's train rows) merge at the end.  1024 pairs cost 8 MFMA + ~130 vector instructions instead of ~340.
__global__ void __launch_bounds__(256) k_knn2_mfma(const unsigned long long* __restrict__ descQ, const int* __restrict__ qoff, const int* __restrict__ nq,
                                                   const unsigned long long* __restrict__ descT, const int* __restrict__ toff, const int* __restrict__ nt,
                                                   int cap, int* __restrict__ idx0, int* __restrict__ dist0, int* __restrict__ idx1,
                                                   int* __restrict__ dist1, uint8_t* __restrict__ ratio_ok) {
    __shared__ unsigned s_top[4][32][2];
    __shared__ uint32_t s_lut[16];
    const int b = (int)blockIdx.y, lane = lane_id(), w = wave_id();
    const int qbase = (int)blockIdx.x * 32;
    const int q0 = qoff ? qoff[b] : 0, t0 = toff ? toff[b] : 0;
    const int nQ = nq[b] - q0, nT = nt[b] - t0;
    unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
    if (qbase < nQ) {                                      // workgroup-uniform
        if (threadIdx.x < 16) {
            const uint32_t bits = (uint32_t)__umul24((uint32_t)threadIdx.x, 0x204081u) & 0x01010101u;      // bit i of the nibble -> bit 0 of byte i
            s_lut[threadIdx.x] = 0x01010101u | ((bits << 8) - bits);                                        // 0 -> 0x01 (+1), 1 -> 0xFF (-1)
        }
        __syncthreads();
        const int r = lane & 31, hoff = 16 * (lane >> 5);
        // 16 bits (k-step s of this lane's 128) -> 16 bytes
        auto expand = [&](const uint32_t raw[4], int s) {
            const uint32_t h = (raw[s >> 1] >> (16 * (s & 1))) & 0xFFFFu;
            v4i_t v; int e[4];
#pragma unroll
            for (int j = 0; j < 4; j++) e[j] = (int)s_lut[(h >> (4 * j)) & 15u];
            __builtin_memcpy(&v, e, 16);
            return v;
        };
        auto load_bits = [&](const unsigned long long* desc, int row, uint32_t raw[4]) {         // rows past the set are masked below; keep the read inside the image's block
            __builtin_memcpy(raw, (const uint8_t*)(desc + 4 * ((size_t)b * cap + imin(row, cap - 1))) + hoff, 16);
        };
        v4i_t qb[8];
        {
            uint32_t raw[4];
            load_bits(descQ, q0 + qbase + r, raw);
#pragma unroll
            for (int s = 0; s < 8; s++) qb[s] = expand(raw, s);
        }
        const int ntiles = (nT + 31) >> 5;
        const int mlane = (256 << 15) + 4 * (lane >> 5);    // (256 - D) << 15 + m = D * -2^15 + (256 << 15) + m: one multiply-add and one add per value
        auto process = [&](const uint32_t raw[4], int tt) {
            v16i_t acc = v16i_zero();
#pragma unroll
            for (int s = 0; s < 8; s++) acc = mfma_i8_32x32x32(expand(raw, s), qb[s], acc);
            const int base = mlane + tt * 32;
            if ((tt + 1) * 32 <= nT) {                     // wave-uniform: every row of the tile is a train row
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const unsigned key = (unsigned)(__mul24(acc[i], -32768) + base + ((i & 3) + 8 * (i >> 2)));
                    k1 = umin32(k1, umax32(k0, key)); k0 = umin32(k0, key);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    unsigned key = (unsigned)(__mul24(acc[i], -32768) + base + ((i & 3) + 8 * (i >> 2)));
                    if ((int)(key & 0xFFFFu) >= nT) key = 0xFFFFFFFFu;      // the low half of the key is the train row (kp_total_cap < 65535, checked at init)
                    k1 = umin32(k1, umax32(k0, key)); k0 = umin32(k0, key);
                }
            }
        };
        // two tile buffers: the load of the next tile is in flight while the matrix cores and the epilogue work on the current one
        uint32_t ta[4], tb[4];
        int tt = w;
        if (tt < ntiles) load_bits(descT, t0 + tt * 32 + r, ta);
        for (; tt + 4 < ntiles; tt += 8) {
            load_bits(descT, t0 + (tt + 4) * 32 + r, tb);
            process(ta, tt);
            if (tt + 8 < ntiles) load_bits(descT, t0 + (tt + 8) * 32 + r, ta);
            process(tb, tt + 4);
        }
        if (tt < ntiles) process(ta, tt);
        // the other half of the train rows of this query
        const unsigned p0 = __shfl_xor(k0, 32), p1 = __shfl_xor(k1, 32);
        const unsigned b0 = umin32(k0, p0), b1 = umin32(umax32(k0, p0), umin32(k1, p1));
        k0 = b0; k1 = b1;
        // the other three waves' train tiles
        if (lane < 32) { s_top[w][lane][0] = k0; s_top[w][lane][1] = k1; }
        __syncthreads();
        if (w == 0 && lane < 32) {
#pragma unroll
            for (int o = 1; o < 4; o++) {
                const unsigned c0 = s_top[o][lane][0], c1 = s_top[o][lane][1];
                const unsigned n1 = umin32(umax32(k0, c0), umin32(k1, c1));
                k0 = umin32(k0, c0); k1 = n1;
            }
        }
    }
    const int q = qbase + lane;
    if (w == 0 && lane < 32 && q < cap) {
        const size_t o = (size_t)b * cap + q;
        if (q >= nQ) { idx0[o] = -1; idx1[o] = -1; dist0[o] = -1; dist1[o] = -1; ratio_ok[o] = 0; }
        else {
            const int i0 = k0 == 0xFFFFFFFFu ? -1 : (int)(k0 & 0xFFFF), i1 = k1 == 0xFFFFFFFFu ? -1 : (int)(k1 & 0xFFFF);
            const int dd0 = i0 < 0 ? -1 : (int)(k0 >> 16), dd1 = i1 < 0 ? -1 : (int)(k1 >> 16);
            idx0[o] = i0; idx1[o] = i1; dist0[o] = dd0; dist1[o] = dd1;
            ratio_ok[o] = (i1 >= 0 && (double)(float)dd0 < (double)(float)dd1 * 0.7) ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The rest of Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1556-1586) after k_knn2: every query (left lapping keypoint) whose best
// match passed the ratio test is triangulated with KannalaBrandt8::TriangulateMatches; depth > 1e-4 accepts.  One thread per query.
// l2r / depth / p3d are indexed by left keypoint, r2l by right keypoint (the loop runs over ascending left index, so the last writer of
// mvRightToLeftMatch[j] is the largest left index: atomicMax).  grid (ceil(cap/256), B).  r2l must be pre-filled with -1.
__global__ void __launch_bounds__(256) k_kb8_stereo(const KeyPointRec* __restrict__ kpsL, const int* __restrict__ monoL, const int* __restrict__ nL,
                                                    const KeyPointRec* __restrict__ kpsR, const int* __restrict__ monoR, int cap,
                                                    const int* __restrict__ idx0, const uint8_t* __restrict__ ratio_ok, KB8StereoParams P,
                                                    int* __restrict__ l2r, int* __restrict__ r2l, float* __restrict__ depth, float* __restrict__ p3d,
                                                    int* __restrict__ nmatches) {
    // Only the queries that passed the ratio test are triangulated (a fifth of the lapping keypoints, scattered): the workgroup first lists them, then
    // its first threads take one listed query each, so that the long fp64 path runs in full waves instead of in a few lanes of every wave.
    __shared__ int s_list[256];
    __shared__ int s_n;
    const int b = (int)blockIdx.y, i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (i < cap) {
        const size_t o = (size_t)b * cap + i;
        l2r[o] = -1; depth[o] = -1.0f;
        p3d[3 * o] = 0.f; p3d[3 * o + 1] = 0.f; p3d[3 * o + 2] = 0.f;
        const int q = i - monoL[b];                              // query row of the kNN (lapping keypoints start at monoLeft)
        if (i < nL[b] && q >= 0 && ratio_ok[(size_t)b * cap + q]) s_list[atomicAdd(&s_n, 1)] = i;
    }
    __syncthreads();
    if ((int)threadIdx.x < s_n) {
        const int i2 = s_list[threadIdx.x];
        const size_t o = (size_t)b * cap + i2;
        const int j = idx0[(size_t)b * cap + (i2 - monoL[b])] + monoR[b];
        const KeyPointRec kl = kpsL[o], kr = kpsR[(size_t)b * cap + j];
        KB8Cam c1, c2;
#pragma unroll
        for (int k = 0; k < 8; k++) { c1.p[k] = P.cam1[k]; c2.p[k] = P.cam2[k]; }
        float r1[3], r2[3], p[3];
        kb8_unproject(c1, kl.x, kl.y, r1);
        kb8_unproject(c2, kr.x, kr.y, r2);
        const float z = kb8_triangulate_matches(c1, c2, r1, r2, kl.x, kl.y, kr.x, kr.y, P.R12, P.t12, pick(P.sigma2, kl.octave), pick(P.sigma2, kr.octave), p);
        if (z > 0.0001f) {
            l2r[o] = j; depth[o] = z;
            p3d[3 * o] = p[0]; p3d[3 * o + 1] = p[1]; p3d[3 * o + 2] = p[2];
            atomicMax(&r2l[(size_t)b * cap + j], i2);
            atomicAdd(&nmatches[b], 1);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:438-529), many map points per launch: for point p with descriptors
// D_0..D_{N-1} (its observations), row i = all N distances d(D_i, D_j) including d(i,i) = 0; median_i = sorted(row i)[0.5*(N-1)];
// best = first i with the smallest median (strict `<`, :511).  One wave per point: a row's distances go into a 257-bin LDS histogram
// (distances are integers in [0, 256]), a wave prefix sum over the bins locates the rank-(N-1)/2 value.  grid (ceil(P/4)), 256 threads.
__global__ void __launch_bounds__(256) k_distinctive(const unsigned long long* __restrict__ desc, const int* __restrict__ start, int P,
                                                     int* __restrict__ best) {
    __shared__ int s_hist[4][320];
    const int lane = lane_id(), w = wave_id();
    const int p = (int)blockIdx.x * 4 + w;
    if (p >= P) return;
    const int s0 = start[p], N = start[p + 1] - s0;
    int* hist = s_hist[w];
#pragma unroll
    for (int t = 0; t < 5; t++) hist[lane * 5 + t] = 0;
    ORBX_WAVE_SYNC();
    if (N <= 0) { if (lane == 0) best[p] = -1; return; }
    const int rank = (N - 1) >> 1;                       // (size_t)(0.5 * (N - 1))
    int bestMedian = 0x7fffffff, bestIdx = 0;
    for (int i = 0; i < N; i++) {
        const unsigned long long* di = desc + 4 * (size_t)(s0 + i);
        const unsigned long long a0 = di[0], a1 = di[1], a2 = di[2], a3 = di[3];
        for (int j = lane; j < N; j += 64) {
            const unsigned long long* dj = desc + 4 * (size_t)(s0 + j);
            const int d = __popcll(a0 ^ dj[0]) + __popcll(a1 ^ dj[1]) + __popcll(a2 ^ dj[2]) + __popcll(a3 ^ dj[3]);
            atomicAdd(&hist[d], 1);
        }
        ORBX_WAVE_SYNC();
        int c[5], mine = 0;
#pragma unroll
        for (int t = 0; t < 5; t++) { c[t] = hist[lane * 5 + t]; mine += c[t]; }
        const int incl = wave_incl_scan(mine);
        int before = incl - mine, median = -1;
        if (before <= rank && rank < incl) {
#pragma unroll
            for (int t = 0; t < 5; t++) { if (median < 0 && rank < before + c[t]) median = lane * 5 + t; before += c[t]; }
        }
        const unsigned long long who = __ballot(median >= 0);
        median = __shfl(median, __ffsll(who) - 1);
        if (median < bestMedian) { bestMedian = median; bestIdx = i; }
#pragma unroll
        for (int t = 0; t < 5; t++) hist[lane * 5 + t] = 0;
        ORBX_WAVE_SYNC();
    }
    if (lane == 0) best[p] = bestIdx;
}

}  // namespace orbx
