// k_search.hip — candidate generation + Hamming evaluation for the guided searches of ORBmatcher:
//   k_grid_build   Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:469-504, :962-978): 64x48 bucket grid as CSR,
//                  items in insertion (= index) order inside a cell
//   k_area_search  Frame::GetFeaturesInArea (src/Frame.cc:859-951) fused with the per-candidate part of
//                  ORBmatcher::SearchByProjection (src/ORBmatcher.cc:45-239 and :1950-2184): one wave64 per query
//                  (map point / last-frame point); emits, in GetFeaturesInArea order (cells ix-major, iy-minor, items in
//                  insertion order), every candidate that passes the level / box / right-coordinate gates together with
//                  its Hamming distance and octave.  The order-dependent acceptance loop ("skip keypoints that already
//                  hold a MapPoint with observations", best / second-best / ratio, rotation histogram) is replayed on the
//                  host over these lists (orbm_search.cpp), because an assignment made for one map point removes that
//                  keypoint from the candidate set of every later one (a true sequential dependency).
//   k_bow_search   inner loops of ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:1045-1323): one wave per
//                  unmatched feature of KF1 against the features of KF2 in the same vocabulary node.
#include "orbx_types.h"
#include "orbx_block.h"
#include "orbx_kernels.h"
#include "kb8_model.h"
#include "sophus_action.h"
#include "glibc_logf_model.h"

namespace orbx {

constexpr int kGridSmallCell = 16;                 // k_grid_build: cells up to this size are ordered by their own thread
constexpr int kGridCols = 64, kGridRows = 48;   // FRAME_GRID_COLS / FRAME_GRID_ROWS, include/Frame.h:44-45

// grid (1), kGridThreads threads.  cell_of: scratch [N] ints.  cell_start: [64*48+1].  Cell id = ix*48 + iy.
// One workgroup because the histogram and the cursors live in LDS; 1024 threads so that a frame's keypoints take one trip per step (the
// kernel is a chain of dependent round trips, not work).
// Batched form (grid = B frames of an extractor's device-resident outputs): n_per_frame != NULL gives N per frame and frame b's arrays sit at
// kps + b * frame_stride, cell_of / cell_items + b * frame_stride, cell_start + b * kGridCellStride.
__global__ void __launch_bounds__(kGridThreads) k_grid_build(const KeyPointRec* __restrict__ kps, int N, GridParams g,
                                                             int* __restrict__ cell_of, int* __restrict__ cell_start,
                                                             int* __restrict__ cell_items, const int* __restrict__ n_per_frame, int frame_stride) {
    if (n_per_frame) {
        const size_t b = blockIdx.x;
        N = n_per_frame[b]; kps += b * (size_t)frame_stride; cell_of += b * (size_t)frame_stride; cell_items += b * (size_t)frame_stride;
        cell_start += b * (size_t)kGridCellStride;
    }
    __shared__ int s_hist[kGridCols * kGridRows];
    __shared__ int s_chunk[256];
    __shared__ unsigned long long s_scan[20];
    const int tid = (int)threadIdx.x;
    constexpr int ncell = kGridCols * kGridRows, NT = kGridThreads, kKeep = 4;
    for (int c = tid; c < ncell; c += NT) s_hist[c] = 0;
    if (tid == 0) s_chunk[0] = 0;                                 // set when a cell is crowded (the ordered chunk placement below then runs)
    __syncthreads();
    int myc[kKeep];                                               // cells of this thread's first keypoints (tid, tid + NT, ..)
#pragma unroll
    for (int k = 0; k < kKeep; k++) myc[k] = -1;
#pragma unroll
    for (int k = 0; k < kKeep; k++) {
        const int i = tid + k * NT;
        if (i < N) {
            const KeyPointRec kp = kps[i];
            const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, g.min_x), g.gw_inv));
            const int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, g.min_y), g.gh_inv));
            if (!(px < 0 || px >= kGridCols || py < 0 || py >= kGridRows)) { myc[k] = px * kGridRows + py; atomicAdd(&s_hist[myc[k]], 1); }
            cell_of[i] = myc[k];
        }
    }
    for (int i = tid + kKeep * NT; i < N; i += NT) {
        const KeyPointRec kp = kps[i];
        const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, g.min_x), g.gw_inv));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, g.min_y), g.gh_inv));
        int c = -1;
        if (!(px < 0 || px >= kGridCols || py < 0 || py >= kGridRows)) { c = px * kGridRows + py; atomicAdd(&s_hist[c], 1); }
        cell_of[i] = c;
    }
    __syncthreads();
    // exclusive scan of the histogram -> cell_start; s_hist becomes the running cursor of each cell
    // (a thread takes ncell / NT = 3 consecutive cells, so one workgroup scan does)
    {
        constexpr int kPer = ncell / NT;
        static_assert(kPer * NT == ncell, "grid cells per thread");
        int loc[kPer], sum = 0, mx = 0;
#pragma unroll
        for (int k = 0; k < kPer; k++) { const int v = s_hist[tid * kPer + k]; loc[k] = sum; sum += v; mx = v > mx ? v : mx; }
        if (mx > kGridSmallCell) s_chunk[0] = 1;                  // (benign race: every writer stores 1)
        unsigned long long tot;
        const int ex = (int)block_excl_scan<unsigned long long>((unsigned long long)sum, &tot, s_scan);
#pragma unroll
        for (int k = 0; k < kPer; k++) { cell_start[tid * kPer + k] = ex + loc[k]; s_hist[tid * kPer + k] = ex + loc[k]; }
        if (tid == 0) cell_start[ncell] = (int)tot;
        __syncthreads();
        if (s_chunk[0] == 0) {
            // the usual frame: no cell holds more than kGridSmallCell keypoints.  Unordered placement through the LDS cursors, then every
            // thread puts the (few) items of each of its cells into index order = the order of the reference's push_back loop (src/Frame.cc:488-503)
#pragma unroll
            for (int k = 0; k < kKeep; k++) if (myc[k] >= 0) cell_items[atomicAdd(&s_hist[myc[k]], 1)] = tid + k * NT;
            for (int i = tid + kKeep * NT; i < N; i += NT) {
                const int c = cell_of[i];
                if (c >= 0) cell_items[atomicAdd(&s_hist[c], 1)] = i;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kPer; k++) {
                const int st = ex + loc[k], cnt = s_hist[tid * kPer + k] - st;
                for (int a2 = 1; a2 < cnt; a2++) {                // insertion sort of <= kGridSmallCell indices
                    const int v = cell_items[st + a2];
                    int j = a2 - 1;
                    while (j >= 0 && cell_items[st + j] > v) { cell_items[st + j + 1] = cell_items[st + j]; j--; }
                    cell_items[st + j + 1] = v;
                }
            }
            return;
        }
    }
    __syncthreads();
    // crowded cells: stable placement, 256 keypoints at a time in index order (the first four waves work, the others keep the barriers company)
    for (int i0 = 0; i0 < N; i0 += 256) {
        const int i = i0 + tid;
        const int c = (tid < 256 && i < N) ? cell_of[i] : -1;
        if (tid < 256) s_chunk[tid] = c;
        __syncthreads();
        int before = 0, after = 0, cur = 0;
        if (c >= 0) {
            for (int t = 0; t < 256; t++) { const int m = (s_chunk[t] == c); before += (t < tid) & m; after += (t > tid) & m; }
            cur = s_hist[c];
        }
        __syncthreads();
        if (c >= 0) {
            cell_items[cur + before] = i;
            if (after == 0) s_hist[c] = cur + before + 1;     // the last lane of this cell in the chunk advances the cursor
        }
        __syncthreads();
    }
}

// level filter (with the reference's bCheckLevels quirk, :908), box test (:944) and right-coordinate gate
// (ORBmatcher.cc:107-117 / :2043-2050) of one keypoint against one query
__device__ __forceinline__ bool area_accept(const AreaQuery& A, const KeyPointRec& k, int idx, bool check_levels,
                                            int gate_right, const float* __restrict__ u_right) {
    if (check_levels) {
        if (k.octave < A.min_level) return false;
        if (A.max_level >= 0 && k.octave > A.max_level) return false;
    }
    if (!(fabsf(__fsub_rn(k.x, A.x)) < A.r && fabsf(__fsub_rn(k.y, A.y)) < A.r)) return false;
    if (gate_right && A.gate == 1) {       // (gate 2 = Fuse's chi-square test, which has no radius test on the right coordinate: src/ORBmatcher.cc:1437-1469)
        const float ur = u_right[idx];
        if (ur > 0 && fabsf(__fsub_rn(A.ur, ur)) > A.r) return false;
    }
    return true;
}
// reprojection gate of ORBmatcher::Fuse (src/ORBmatcher.cc:1437-1469): e2 * invSigma2[octave] > 7.8 (stereo keypoint, 3 dof)
// or > 5.99 (monocular, 2 dof), float product compared as double like the reference's float-vs-double-literal comparison
__device__ __forceinline__ bool chi2_accept(const AreaQuery& A, const KeyPointRec& k, float ur, const GridParams& g) {
    const float ex = __fsub_rn(A.x, k.x), ey = __fsub_rn(A.y, k.y);
    float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
    const float inv = g.inv_sigma2[k.octave & (kMaxLevels - 1)];
    if (ur >= 0) {
        const float er = __fsub_rn(A.ur, ur);
        e2 = __fadd_rn(e2, __fmul_rn(er, er));
        return !((double)__fmul_rn(e2, inv) > 7.8);
    }
    return !((double)__fmul_rn(e2, inv) > 5.99);
}

// One wave per query.  grid (ceil(Q / kAreaWaves)), 64 * kAreaWaves threads.  Lane l of a chunk owns window cell l (cells enumerated ix-major,
// iy-minor like the reference's loops): pass 0 counts, the workgroup reserves the spans of its queries in the entry pool with ONE atomicAdd
// (thousands of same-address atomics, one per query, were most of this kernel's time), pass 1 re-enumerates and writes each lane's
// candidates at its ordered offset (wave prefix sum).
// entries: int2 per candidate {idx, dist | octave << 16}; q_start/q_count: CSR; pool_counter: running allocation.
__global__ void __launch_bounds__(64 * kAreaWaves) k_area_search(const AreaQuery* __restrict__ queries, const unsigned long long* __restrict__ qdesc, int Q,
                                                     const KeyPointRec* __restrict__ kps, const float* __restrict__ u_right,
                                                     const unsigned long long* __restrict__ fdesc, GridParams g,
                                                     const int* __restrict__ cell_start, const int* __restrict__ cell_items,
                                                     int gate_right, int* __restrict__ pool_counter, int pool_cap,
                                                     int* __restrict__ q_start, int* __restrict__ q_count, int2* __restrict__ entries, int frame_stride) {
    __shared__ int s_cnt[kAreaWaves], s_base;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int q = (int)blockIdx.x * kAreaWaves + wave;
    if (frame_stride > 0) {
        // batched form: blockIdx.y = frame; every frame has its own Q queries against its own keypoints / grid; the query descriptors (map
        // points) and the entry pool are shared
        const size_t b = blockIdx.y;
        queries += b * (size_t)Q; q_start += b * (size_t)Q; q_count += b * (size_t)Q;
        kps += b * (size_t)frame_stride; u_right += b * (size_t)frame_stride; fdesc += 4 * b * (size_t)frame_stride;
        cell_start += b * (size_t)kGridCellStride; cell_items += b * (size_t)frame_stride;
    }
    AreaQuery A{};
    if (q < Q) A = queries[q];
    int cnt_total = 0, start = 0;
    // window of grid cells, src/Frame.cc:877-903
    const int nMinX = imax(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(A.x, g.min_x), A.r), g.gw_inv)));
    const int nMaxX = imin(kGridCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(A.x, g.min_x), A.r), g.gw_inv)));
    const int nMinY = imax(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(A.y, g.min_y), A.r), g.gh_inv)));
    const int nMaxY = imin(kGridRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(A.y, g.min_y), A.r), g.gh_inv)));
    const bool window_ok = q < Q && A.active && !(nMinX >= kGridCols || nMaxX < 0 || nMinY >= kGridRows || nMaxY < 0) && nMaxX >= nMinX && nMaxY >= nMinY;
    const int ny = nMaxY - nMinY + 1, ncw = window_ok ? (nMaxX - nMinX + 1) * ny : 0;
    const bool check_levels = (A.min_level > 0) || (A.max_level >= 0);
    unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    if (window_ok) { const unsigned long long* dq = qdesc + 4 * (size_t)q; d0 = dq[0]; d1 = dq[1]; d2 = dq[2]; d3 = dq[3]; }
    {
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1 && cnt_total == 0) break;
            int base = 0;
            for (int c0 = 0; c0 < ncw; c0 += 64) {
                const int c = c0 + lane;
                int s = 0, e = 0;
                if (c < ncw) {
                    const int ix = nMinX + c / ny, iy = nMinY + c % ny;
                    s = cell_start[ix * kGridRows + iy]; e = cell_start[ix * kGridRows + iy + 1];
                }
                int cnt = 0;
                for (int j = s; j < e; j++) {
                    const int idx = cell_items[j];
                    const KeyPointRec k = kps[idx];
                    cnt += (area_accept(A, k, idx, check_levels, gate_right, u_right) && (A.gate != 2 || chi2_accept(A, k, u_right[idx], g))) ? 1 : 0;
                }
                const int incl = wave_incl_scan(cnt);
                if (pass == 1 && start >= 0) {
                    int pos = start + base + incl - cnt;
                    for (int j = s; j < e; j++) {
                        const int idx = cell_items[j];
                        const KeyPointRec k = kps[idx];
                        if (!area_accept(A, k, idx, check_levels, gate_right, u_right)) continue;
                        if (A.gate == 2 && !chi2_accept(A, k, u_right[idx], g)) continue;
                        const unsigned long long* df = fdesc + 4 * (size_t)idx;
                        const int dist = __popcll(d0 ^ df[0]) + __popcll(d1 ^ df[1]) + __popcll(d2 ^ df[2]) + __popcll(d3 ^ df[3]);
                        int2 ent; ent.x = idx; ent.y = dist | (k.octave << 16);
                        entries[pos++] = ent;
                    }
                }
                base += __shfl(incl, 63);
            }
            if (pass == 0) {
                cnt_total = base;
                // one reservation per workgroup: the waves' spans follow each other in wave order
                if (lane == 0) s_cnt[wave] = cnt_total;
                __syncthreads();
                if (threadIdx.x == 0) {
                    int tot = 0;
                    for (int w = 0; w < kAreaWaves; w++) tot += s_cnt[w];
                    s_base = tot > 0 ? atomicAdd(pool_counter, tot) : 0;
                }
                __syncthreads();
                int off = s_base;
                for (int w = 0; w < wave; w++) off += s_cnt[w];
                start = (off + cnt_total <= pool_cap) ? off : -1;   // -1: pool overflow, the host retries with a bigger pool
            }
        }
    }
    if (lane == 0 && q < Q) { q_start[q] = start < 0 ? 0 : start; q_count[q] = cnt_total; }
}

// Frame::isInFrustum (src/Frame.cc:667-773, one camera) + MapPoint::PredictScale (src/MapPoint.cc:688-731) for M map points, one thread each,
// in the reference's fp32 operation order (Eigen >= 3.3 sums a 3-term product coefficient as a0 + (a1 + a2), sophus_action.h; no fused multiply-adds).  Writes the tracking
// fields the reference stores in the MapPoint (mbTrackInView, mTrackProjX / Y / XR, mTrackDepth, mnTrackScaleLevel, mTrackViewCos) and,
// when `queries` is given, the window query of ORBmatcher::SearchByProjection(Frame, MapPoints) for that point (src/ORBmatcher.cc:53-82).
// Batched form: Fbatch != NULL -> blockIdx.y = frame, frame b uses Fbatch[b] and writes at offsets b * M (track: b * 5 * M).
// F.rig_mode (Frame::isInFrustumChecks, src/Frame.cc:1592-1650, one camera of a two-camera rig; the caller passes that camera's mR, mt, twc
// and parameters): nothing is stored unless the point passes every test, and the level of a rejected point is -1 (:756-757).
// (the body takes the frame's parameters by REFERENCE - the kernel argument itself, or the batch's record in global memory: copying one over the
// other (`F = Fbatch[b]`) made the struct a private variable, 240 bytes of scratch per thread)
__device__ __forceinline__ void frustum_body(const FrustumParams& F, int i, int M, const float* __restrict__ pos, const float* __restrict__ normal,
                                             const float* __restrict__ min_dist, const float* __restrict__ max_dist, const uint8_t* __restrict__ is_bad,
                                             uint8_t* __restrict__ in_view, float* __restrict__ track, int* __restrict__ scale_level, AreaQuery* __restrict__ queries) {
    const float P0 = pos[3 * i], P1 = pos[3 * i + 1], P2 = pos[3 * i + 2];
    bool ok = true;
    float u = -1.0f, v = -1.0f, xr = 0.0f, depth = 0.0f, vcos = 0.0f; int lvl = 0;
    // Pc = mRcw * P + mtcw
    // (the reference works with the MATRIX here, src/Frame.cc:685; Eigen's 3-term sums are a0 + (a1 + a2), sophus_action.h)
    float Pc[3];
    eig_rt3(F.Rcw, F.tcw, P0, P1, P2, Pc);
    const float x = Pc[0], y = Pc[1], z = Pc[2];
    const float Pc_dist = sqrtf(eig_dot3(x, y, z, x, y, z));
    const float invz = __fdiv_rn(1.0f, z);
    if (z < 0.0f) ok = false;
    float uu = 0.f, vv = 0.f;
    if (ok) {
        if (F.kb8) { KB8Cam c; for (int k = 0; k < 8; k++) c.p[k] = F.cam[k]; const float pc[3] = {x, y, z}; float uv[2]; kb8_project(c, pc, uv); uu = uv[0]; vv = uv[1]; }
        else {                                                        // Pinhole::project, src/CameraModels/Pinhole.cpp:61-68
            uu = __fadd_rn(__fdiv_rn(__fmul_rn(F.cam[0], x), z), F.cam[2]);
            vv = __fadd_rn(__fdiv_rn(__fmul_rn(F.cam[1], y), z), F.cam[3]);
        }
        if (uu < F.min_x || uu > F.max_x) ok = false;
        if (ok && (vv < F.min_y || vv > F.max_y)) ok = false;
    }
    if (ok) {
        if (!F.rig_mode) { u = uu; v = vv; }                          // mTrackProjX / Y are set before the distance tests (:705-706)
        const float maxDistance = __fmul_rn(1.2f, max_dist[i]), minDistance = __fmul_rn(0.8f, min_dist[i]);     // MapPoint.cc:658-671
        const float o0 = __fsub_rn(P0, F.Ow[0]), o1 = __fsub_rn(P1, F.Ow[1]), o2 = __fsub_rn(P2, F.Ow[2]);
        const float dist = sqrtf(eig_dot3(o0, o1, o2, o0, o1, o2));
        if (dist < minDistance || dist > maxDistance) ok = false;
        if (ok) {
            const float dot = eig_dot3(o0, o1, o2, normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
            vcos = __fdiv_rn(dot, dist);
            if (vcos < F.cos_limit) ok = false;
        }
        if (ok) {
            // MapPoint::PredictScale (src/MapPoint.cc:714-731): ceil(log(ratio) / mfLogScaleFactor) in FLOAT - std::log(float) = glibc's logf
            // (glibc_logf_model.h, checked against the live libm for every positive float), float division, ceil(float)
            const float ratio = __fdiv_rn(max_dist[i], dist);
            int n = (int)ceilf(__fdiv_rn(glibc_logf_model<false>(ratio), F.log_scale_factor));
            if (n < 0) n = 0; else if (n >= F.nlevels) n = F.nlevels - 1;
            lvl = n;
            u = uu; v = vv;
            xr = __fsub_rn(uu, __fmul_rn(F.mbf, invz));
            depth = Pc_dist;
        }
    }
    in_view[i] = ok ? 1 : 0;
    track[i] = u; track[M + i] = v; track[2 * (size_t)M + i] = xr; track[3 * (size_t)M + i] = depth; track[4 * (size_t)M + i] = vcos;
    scale_level[i] = ok ? lvl : (F.rig_mode ? -1 : 0);
    if (queries) {
        AreaQuery q; q.x = 0; q.y = 0; q.r = 0; q.ur = 0; q.min_level = 0; q.max_level = 0; q.active = 0; q.gate = 0;
        if (ok && !(F.far_points && depth > F.th_far) && !(is_bad && is_bad[i])) {
            float r = (double)vcos > 0.998 ? 2.5f : 4.0f;             // RadiusByViewingCos, src/ORBmatcher.cc:242-249
            if (F.th != 1.0f) r = __fmul_rn(r, F.th);
            q.x = u; q.y = v; q.r = __fmul_rn(r, pick(F.scale_factors, lvl)); q.ur = xr;
            q.min_level = lvl - 1; q.max_level = lvl; q.active = 1; q.gate = 1;
        }
        queries[i] = q;
    }
}

__global__ void __launch_bounds__(256) k_frustum(FrustumParams F, int M, const float* __restrict__ pos, const float* __restrict__ normal,
                                                 const float* __restrict__ min_dist, const float* __restrict__ max_dist, const uint8_t* __restrict__ is_bad,
                                                 uint8_t* __restrict__ in_view, float* __restrict__ track /* [6][M]: x, y, xr, depth, cos, - */,
                                                 int* __restrict__ scale_level, AreaQuery* __restrict__ queries, int* __restrict__ zero4,
                                                 const FrustumParams* __restrict__ Fbatch) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (zero4 && i < 4 && blockIdx.y == 0) zero4[i] = 0;          // the pool counter of the window search that follows (saves a fill launch)
    if (i >= M) return;
    if (Fbatch) {
        const size_t b = blockIdx.y;
        frustum_body(Fbatch[b], i, M, pos, normal, min_dist, max_dist, is_bad, in_view + b * (size_t)M, track + 5 * b * (size_t)M, scale_level + b * (size_t)M,
                     queries ? queries + b * (size_t)M : queries);
    } else frustum_body(F, i, M, pos, normal, min_dist, max_dist, is_bad, in_view, track, scale_level, queries);
}

// The head of the projection-type searches of ORBmatcher (SearchByProjection(Frame, LastFrame) src/ORBmatcher.cc:1993-2010, (Frame, KeyFrame)
// :2228-2256, (KeyFrame, Sim3, ...) :525-560 / :640-690, Fuse :1388-1430 / :1590-1625, SearchBySim3 :1745-1790 / :1830-1875): transform the map
// point, depth test, projection, image test, distance range, viewing angle - one thread per point, the reference's fp32 operation order
// (Sophus' quaternion action for the transform, Eigen's a0 + (a1 + a2) for norms and dot products; no fused multiply-adds).  Which tests run and how the projection is written differ between the reference's
// methods; P says which.  Outputs: valid, u, v, ur = u - bf / z, 1 / z, dist (the argument of MapPoint::PredictScale, which stays with the
// caller's MapPoint).  skip[i] != 0: the caller's own tests (bad, already matched, ...) have rejected the point.
__global__ void __launch_bounds__(256) k_project_points(ProjectParams P, int M, const float* __restrict__ pos, const float* __restrict__ normal,
                                                        const float* __restrict__ min_inv, const float* __restrict__ max_inv, const uint8_t* __restrict__ skip,
                                                        uint8_t* __restrict__ valid, float* __restrict__ out /* [5][M]: u, v, ur, 1/z, dist */, int debug_flags) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= M) return;
    bool ok = !(skip && skip[i]);
    float u = 0.f, v = 0.f, ur = 0.f, invz = 0.f, dist = 0.f;
    if (ok) {
        const float P0 = pos[3 * i], P1 = pos[3 * i + 1], P2 = pos[3 * i + 2];
        float pc[3];
        if (debug_flags & 16) se3_act_matrix_form(P.q, P.t, P0, P1, P2, pc);       // test switch: round 3's R * p + t
        else se3_act(P.q, P.t, P0, P1, P2, pc);                       // Tcw * p3Dw: Sophus' quaternion action (so3.hpp:357-367, se3.hpp:321-324)
        if (P.second == 1) sim3_act(P.q2, P.s2, P.t2, pc[0], pc[1], pc[2], pc);       // S21 * p3Dc1 (rxso3.hpp:265-273, sim3.hpp:226-229)
        else if (P.second == 2) se3_act(P.q2, P.t2, pc[0], pc[1], pc[2], pc);         // GetRelativePoseTrl() * x3Dc
        const float x = pc[0], y = pc[1], z = pc[2];
        invz = __fdiv_rn(1.0f, z);
        if (P.depth_test == 1 && z < 0.0f) ok = false;
        if (P.depth_test == 2 && invz < 0.0f) ok = false;
        if (ok) {
            if (P.camera_type == 1) { KB8Cam c; for (int k = 0; k < 8; k++) c.p[k] = P.cam[k]; const float pc[3] = {x, y, z}; float uv[2]; kb8_project(c, pc, uv); u = uv[0]; v = uv[1]; }
            else if (P.inline_pinhole) {                              // x = X * invz; u = fx * x + cx (the hand-written form, e.g. :656-660)
                u = __fadd_rn(__fmul_rn(P.cam[0], __fmul_rn(x, invz)), P.cam[2]);
                v = __fadd_rn(__fmul_rn(P.cam[1], __fmul_rn(y, invz)), P.cam[3]);
            } else {                                                  // Pinhole::project, src/CameraModels/Pinhole.cpp:61-68
                u = __fadd_rn(__fdiv_rn(__fmul_rn(P.cam[0], x), z), P.cam[2]);
                v = __fadd_rn(__fdiv_rn(__fmul_rn(P.cam[1], y), z), P.cam[3]);
            }
            if (P.bounds_mode == 0) { if (u < P.min_x || u > P.max_x || v < P.min_y || v > P.max_y) ok = false; }        // Frame: :2003-2006
            else if (P.bounds_mode == 1 && !(u >= P.min_x && u < P.max_x && v >= P.min_y && v < P.max_y)) ok = false;      // KeyFrame::IsInImage; 2: no image test
        }
        if (ok) {
            ur = __fsub_rn(u, __fmul_rn(P.bf, invz));
            const float o0 = __fsub_rn(P0, P.Ow[0]), o1 = __fsub_rn(P1, P.Ow[1]), o2 = __fsub_rn(P2, P.Ow[2]);
            if (P.dist_mode == 0) dist = sqrtf(eig_dot3(o0, o1, o2, o0, o1, o2));        // PO.norm(): Eigen reduction, a0 + (a1 + a2)
            else dist = sqrtf(eig_dot3(x, y, z, x, y, z));
            if (P.distance_test && (dist < min_inv[i] || dist > max_inv[i])) ok = false;
            if (ok && P.angle_test) {                                 // PO.dot(Pn) < 0.5 * dist3D
                const float dot = eig_dot3(o0, o1, o2, normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
                if (dot < __fmul_rn(0.5f, dist)) ok = false;
            }
        }
    }
    valid[i] = ok ? 1 : 0;
    const size_t Ms = (size_t)M;
    out[i] = u; out[Ms + i] = v; out[2 * Ms + i] = ur; out[3 * Ms + i] = invz; out[4 * Ms + i] = dist;
}

// The head of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1970-2023) for a batch of frames: map point
// i of frame b's last frame -> camera (Tcw * x3Dw), 1 / z < 0 rejects, projection, the Frame's image test, radius = th * mvScaleFactors[octave
// of the LAST frame's keypoint], the level window by bForward / bBackward, ur = u - mbf / z for the right-coordinate gate.  grid (capL / 256, B).
// KEYFRAME = true: the head of SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (relocalisation, :2213-2246) instead: no depth
// test, the distance |x3Dw - Ow| must lie in [0.8 mfMinDistance, 1.2 mfMaxDistance] (MapPoint.cc:658-671), the level is MapPoint::PredictScale
// (dist3D, &CurrentFrame) (MapPoint.cc:714-731: glibc's logf, float division, ceil), window [level - 1, level + 1], no right-coordinate gate.
template <bool KEYFRAME>
__device__ __forceinline__ void projection_queries_body(const FrustumParams* __restrict__ Fb, int capL, const int* __restrict__ n_last, const float* __restrict__ pos,
                                                        const uint8_t* __restrict__ valid, const int* __restrict__ octave, const float* __restrict__ min_dist,
                                                        const float* __restrict__ max_dist, AreaQuery* __restrict__ queries, int* __restrict__ zero4) {
    const size_t b = blockIdx.y;
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (zero4 && i < 4 && b == 0) zero4[i] = 0;
    if (i >= capL) return;
    const FrustumParams& F = Fb[b];                                // by reference: a private copy would live in scratch memory
    const size_t o = b * (size_t)capL + i;
    AreaQuery q; q.x = 0; q.y = 0; q.r = 0; q.ur = 0; q.min_level = 0; q.max_level = 0; q.active = 0; q.gate = 0;
    if (i < n_last[b] && valid[o]) {
        const float P0 = pos[3 * o], P1 = pos[3 * o + 1], P2 = pos[3 * o + 2];
        float pc[3];
        if (F.debug_flags & 16) se3_act_matrix_form(F.qcw, F.tcw, P0, P1, P2, pc);  // test switch: round 3's R * p + t
        else se3_act(F.qcw, F.tcw, P0, P1, P2, pc);                   // x3Dc = Tcw * x3Dw (:1987, :2224): Sophus' quaternion action, not mRcw * p
        const float x = pc[0], y = pc[1], z = pc[2];
        const float invz = __fdiv_rn(1.0f, z);
        const int oct = KEYFRAME ? 0 : octave[o];
        if (KEYFRAME || (!(invz < 0.0f) && oct >= 0 && oct < F.nlevels)) {
            float u, v;
            if (F.kb8) { KB8Cam c; for (int k = 0; k < 8; k++) c.p[k] = F.cam[k]; const float pcc[3] = {x, y, z}; float uv[2]; kb8_project(c, pcc, uv); u = uv[0]; v = uv[1]; }
            else { u = __fadd_rn(__fdiv_rn(__fmul_rn(F.cam[0], x), z), F.cam[2]); v = __fadd_rn(__fdiv_rn(__fmul_rn(F.cam[1], y), z), F.cam[3]); }
            if (!(u < F.min_x || u > F.max_x || v < F.min_y || v > F.max_y)) {
                if (KEYFRAME) {
                    const float o0 = __fsub_rn(P0, F.Ow[0]), o1 = __fsub_rn(P1, F.Ow[1]), o2 = __fsub_rn(P2, F.Ow[2]);
                    const float dist3D = sqrtf(eig_dot3(o0, o1, o2, o0, o1, o2));                              // PO.norm()
                    const float maxDistance = __fmul_rn(1.2f, max_dist[o]), minDistance = __fmul_rn(0.8f, min_dist[o]);
                    if (!(dist3D < minDistance || dist3D > maxDistance)) {
                        const float ratio = __fdiv_rn(max_dist[o], dist3D);
                        int n = (int)ceilf(__fdiv_rn(glibc_logf_model<false>(ratio), F.log_scale_factor));
                        if (n < 0) n = 0; else if (n >= F.nlevels) n = F.nlevels - 1;
                        q.x = u; q.y = v; q.r = __fmul_rn(F.th, pick(F.scale_factors, n)); q.ur = 0.0f;
                        q.min_level = n - 1; q.max_level = n + 1; q.active = 1; q.gate = 0;
                    }
                } else {
                    q.x = u; q.y = v; q.r = __fmul_rn(F.th, pick(F.scale_factors, oct)); q.ur = __fsub_rn(u, __fmul_rn(F.mbf, invz));
                    if (F.forward) { q.min_level = oct; q.max_level = -1; }
                    else if (F.backward) { q.min_level = 0; q.max_level = oct; }
                    else { q.min_level = oct - 1; q.max_level = oct + 1; }
                    q.active = 1; q.gate = 1;
                }
            }
        }
    }
    queries[o] = q;
}
__global__ void __launch_bounds__(256) k_lastframe_queries(const FrustumParams* __restrict__ Fb, int capL, const int* __restrict__ n_last, const float* __restrict__ pos,
                                                           const uint8_t* __restrict__ valid, const int* __restrict__ octave, AreaQuery* __restrict__ queries,
                                                           int* __restrict__ zero4) {
    projection_queries_body<false>(Fb, capL, n_last, pos, valid, octave, nullptr, nullptr, queries, zero4);
}
__global__ void __launch_bounds__(256) k_keyframe_queries(const FrustumParams* __restrict__ Fb, int capL, const int* __restrict__ n_kf, const float* __restrict__ pos,
                                                          const uint8_t* __restrict__ valid, const float* __restrict__ min_dist, const float* __restrict__ max_dist,
                                                          AreaQuery* __restrict__ queries, int* __restrict__ zero4) {
    projection_queries_body<true>(Fb, capL, n_kf, pos, valid, nullptr, min_dist, max_dist, queries, zero4);
}

// The window search of a BATCH of frames, one THREAD per query (k_area_search spends a wave on a query: right for one frame's few thousand
// queries, which must finish in microseconds; a batch has hundreds of thousands and wants throughput).  A thread walks its window cells in the
// reference's order (ix-major, iy-minor, items in insertion order), counts, the workgroup reserves its span of the entry pool with one
// atomicAdd, and a second walk writes {idx, dist | octave << 16} at the thread's offset.  Same gates as k_area_search.  blockIdx.y = frame.
// On pool overflow nothing is written and the counts are 0 (the host sees the counter, enlarges the pool and repeats).
__global__ void __launch_bounds__(256) k_area_search_threads(const AreaQuery* __restrict__ queries, const unsigned long long* __restrict__ qdesc, int Q,
                                                             const KeyPointRec* __restrict__ kps, const float* __restrict__ u_right,
                                                             const unsigned long long* __restrict__ fdesc, GridParams g, const int* __restrict__ cell_start,
                                                             const int* __restrict__ cell_items, int gate_right, int* __restrict__ pool_counter, int pool_cap,
                                                             int* __restrict__ q_start, int* __restrict__ q_count, int2* __restrict__ entries, int frame_stride,
                                                             int qdesc_per_frame) {
    __shared__ int s_scan[20];
    __shared__ int s_base;
    // the first kAreaKeep accepted keypoints of every query (index | octave << 16), [slot][thread]: the second pass - Hamming distances and the
    // writes - then reads them back instead of walking the window, its cells, keypoint records and gates again (round 4; most queries accept fewer)
    constexpr int kAreaKeep = 12;
    __shared__ uint32_t s_keep[kAreaKeep * 256];
    const size_t b = blockIdx.y;
    const int q = (int)(blockIdx.x * 256 + threadIdx.x);
    queries += b * (size_t)Q; q_start += b * (size_t)Q; q_count += b * (size_t)Q;
    if (qdesc_per_frame) qdesc += 4 * b * (size_t)Q;               // every frame brings its own query descriptors (the map points of ITS last frame)
    kps += b * (size_t)frame_stride; u_right += b * (size_t)frame_stride; fdesc += 4 * b * (size_t)frame_stride;
    cell_start += b * (size_t)kGridCellStride; cell_items += b * (size_t)frame_stride;
    AreaQuery A{};
    if (q < Q) A = queries[q];
    const int nMinX = imax(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(A.x, g.min_x), A.r), g.gw_inv)));
    const int nMaxX = imin(kGridCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(A.x, g.min_x), A.r), g.gw_inv)));
    const int nMinY = imax(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(A.y, g.min_y), A.r), g.gh_inv)));
    const int nMaxY = imin(kGridRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(A.y, g.min_y), A.r), g.gh_inv)));
    const bool window_ok = q < Q && A.active && !(nMinX >= kGridCols || nMaxX < 0 || nMinY >= kGridRows || nMaxY < 0) && nMaxX >= nMinX && nMaxY >= nMinY;
    const bool check_levels = (A.min_level > 0) || (A.max_level >= 0);
    int cnt = 0;
    if (window_ok) {
        for (int ix = nMinX; ix <= nMaxX; ix++) {
            const int s = cell_start[ix * kGridRows + nMinY], e = cell_start[ix * kGridRows + nMaxY + 1];    // the cells of one column are consecutive
            for (int j = s; j < e; j++) {
                const int idx = cell_items[j];
                const KeyPointRec k = kps[idx];
                if (area_accept(A, k, idx, check_levels, gate_right, u_right) && (A.gate != 2 || chi2_accept(A, k, u_right[idx], g))) {
                    if (cnt < kAreaKeep) s_keep[cnt * 256 + (int)threadIdx.x] = (uint32_t)idx | ((uint32_t)k.octave << 16);
                    cnt++;
                }
            }
        }
    }
    int total;
    const int excl = block_excl_scan<int>(cnt, &total, s_scan);
    if (threadIdx.x == 0) s_base = total > 0 ? atomicAdd(pool_counter, total) : 0;
    __syncthreads();
    const bool fits = s_base + total <= pool_cap;
    const int start = s_base + excl;
    if (window_ok && cnt > 0 && fits) {
        const unsigned long long* dq = qdesc + 4 * (size_t)q;
        const unsigned long long d0 = dq[0], d1 = dq[1], d2 = dq[2], d3 = dq[3];
        int pos = start;
        if (cnt <= kAreaKeep) {
            for (int i = 0; i < cnt; i++) {
                const uint32_t e = s_keep[i * 256 + (int)threadIdx.x];
                const int idx = (int)(e & 0xFFFFu);
                const unsigned long long* df = fdesc + 4 * (size_t)idx;
                const int dist = __popcll(d0 ^ df[0]) + __popcll(d1 ^ df[1]) + __popcll(d2 ^ df[2]) + __popcll(d3 ^ df[3]);
                int2 ent; ent.x = idx; ent.y = dist | (int)(e & 0xFFFF0000u);
                entries[pos++] = ent;
            }
        } else
        for (int ix = nMinX; ix <= nMaxX; ix++) {
            const int s = cell_start[ix * kGridRows + nMinY], e = cell_start[ix * kGridRows + nMaxY + 1];
            for (int j = s; j < e; j++) {
                const int idx = cell_items[j];
                const KeyPointRec k = kps[idx];
                if (!area_accept(A, k, idx, check_levels, gate_right, u_right)) continue;
                if (A.gate == 2 && !chi2_accept(A, k, u_right[idx], g)) continue;
                const unsigned long long* df = fdesc + 4 * (size_t)idx;
                const int dist = __popcll(d0 ^ df[0]) + __popcll(d1 ^ df[1]) + __popcll(d2 ^ df[2]) + __popcll(d3 ^ df[3]);
                int2 ent; ent.x = idx; ent.y = dist | (k.octave << 16);
                entries[pos++] = ent;
            }
        }
    }
    if (q < Q) { q_start[q] = fits ? start : 0; q_count[q] = fits ? cnt : 0; }
}

// ORBmatcher::SearchByProjection(Frame, MapPoints) accept loop (src/ORBmatcher.cc:62-166) on the device, one wave per frame of a batch.
// The loop is sequential over the map points: a keypoint that has received a map point WITH observations is skipped by every later point.
// Conflicts are rare, so the wave works on 64 consecutive map points at a time, one per lane, optimistically:
//   1. every pending lane walks its candidates against the current occupancy: best / second best exactly as the strict `<` tests of the
//      reference leave them (the FIRST candidate with the smallest distance; the second best is the smallest among the others), decision;
//   2. lanes whose decision occupies a keypoint publish the claim (LDS, minimum lane per keypoint);
//   3. a lane is DIRTY if an earlier lane of this round claims one of its free candidates - its decision may not stand;
//   4. the lanes before the first dirty one commit (their decisions are what the sequential loop produces: nothing earlier touches their
//      candidates), the others repeat from 1 with the new occupancy.  The first pending lane is never dirty, so every round commits.
// F.mvpMapPoints[idx] = pMP without observations does not occupy; a later point may overwrite it: assignments are merged with atomicMax on
// the map point index (the sequential last writer is the largest index).
// occupied0: [B][cap] bytes or NULL; has_obs: [M] or NULL (all observed).  assigned: [B][cap], -1 = untouched; nmatches: [B].
// dynamic LDS: occupancy bitmap ((cap + 31) / 32 words) | claiming lane per keypoint (cap words) | LASTFRAME: one accept event per query (M words).
// LASTFRAME = the accept loop of SearchByProjection(CurrentFrame, LastFrame) (src/ORBmatcher.cc:2025-2150): the best candidate alone decides
// (bestDist <= TH_HIGH, no ratio test), has_obs / the query arrays are per frame, and every accepted pair goes into the rotation histogram
// (:2118-2126, ComputeThreeMaxima :2335-2377): after the loop the pairs outside the three fullest bins are taken back (assigned = -2 = reset to
// NULL, nmatches--) - per accept EVENT, as the reference's rotHist lists are (a keypoint that was given twice has two entries).
template <bool LASTFRAME>
__device__ __forceinline__ void local_accept_body(int M, int cap, const int* __restrict__ n_per_frame, const int* __restrict__ q_start,
                                                  const int* __restrict__ q_count, const int2* __restrict__ entries, const uint8_t* __restrict__ occupied0,
                                                  const uint8_t* __restrict__ has_obs, float nnratio, int th_high, int* __restrict__ assigned,
                                                  int* __restrict__ nmatches, const float* __restrict__ last_angle, const KeyPointRec* __restrict__ cur_kps,
                                                  int check_ori) {
    int* events = nullptr;
    ORBX_DYN_SMEM(smem);
    const int nwords = (cap + 31) / 32;
    uint32_t* s_occ = (uint32_t*)smem;
    unsigned* s_claim = (unsigned*)(smem + 4 * (size_t)nwords);
    __shared__ int s_hist[32];
    const int lane = lane_id();
    const size_t b = blockIdx.x;
    const int N = n_per_frame[b];
    q_start += b * (size_t)M; q_count += b * (size_t)M; assigned += b * (size_t)cap;
    if (LASTFRAME) {
        if (has_obs) has_obs += b * (size_t)M;
        last_angle += b * (size_t)M; cur_kps += b * (size_t)cap;
        events = (int*)(s_claim + cap);                               // in LDS: written and read by different lanes of this wave, in program order
        if (lane < 32) s_hist[lane] = 0;
    }
    for (int w = lane; w < nwords; w += 64) {
        uint32_t bits = 0;
        if (occupied0) for (int k = 0; k < 32; k++) { const int i = 32 * w + k; if (i < N && occupied0[b * (size_t)cap + i]) bits |= 1u << k; }
        s_occ[w] = bits;
    }
    for (int i = lane; i < cap; i += 64) { assigned[i] = -1; s_claim[i] = 0xFFu; }
    ORBX_WAVE_SYNC();
    int nm = 0;
    for (int i0 = 0; i0 < M; i0 += 64) {
        const int qi = i0 + lane;
        const int cnt = qi < M ? q_count[qi] : 0, st = qi < M ? q_start[qi] : 0;
        const bool obs = qi < M && (!has_obs || has_obs[qi]);
        bool pending = cnt > 0;
        if (LASTFRAME && qi < M) events[qi] = -1;
        while (__ballot(pending) != 0ull) {
            // 1. decision against the current occupancy
            // keys (dist << 16 | position), packed dist | level << 16 of both, index of the best.  16 bits hold every position and keypoint index: a
            // query's candidates are distinct keypoints of one frame, and a frame has fewer than 65535 (kp_total_cap, refused in orbx_api.cpp: configure)
            unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu; int e1 = 0, e2 = 0, idx1 = -1;
            if (pending) {
                for (int k = 0; k < cnt; k++) {
                    const int2 e = entries[st + k];
                    if ((s_occ[e.x >> 5] >> (e.x & 31)) & 1u) continue;
                    const unsigned key = ((unsigned)(e.y & 0xFFFF) << 16) | (unsigned)k;
                    if (key < k1) { k2 = k1; e2 = e1; k1 = key; e1 = e.y; idx1 = e.x; } else if (key < k2) { k2 = key; e2 = e.y; }
                }
            }
            bool accept = false;
            if (pending && k1 != 0xFFFFFFFFu) {
                const int bestDist = e1 & 0xFFFF, bestLevel = e1 >> 16;
                int bestDist2 = 256, bestLevel2 = -1;
                if (k2 != 0xFFFFFFFFu) { bestDist2 = e2 & 0xFFFF; bestLevel2 = e2 >> 16; }
                // :146-166  bestDist <= TH_HIGH, and not (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2);  LastFrame (:2103): bestDist <= TH_HIGH
                accept = bestDist <= th_high && (LASTFRAME || !(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2)));
            }
            // 2. claims that occupy
            const bool claims = accept && obs;
            if (claims) atomicMin(&s_claim[idx1], (unsigned)lane);
            ORBX_WAVE_SYNC();
            // 3. dirty: an earlier lane claims one of my free candidates
            bool dirty = false;
            if (pending) {
                for (int k = 0; k < cnt; k++) {
                    const int idx = entries[st + k].x;
                    if (s_claim[idx] < (unsigned)lane) { dirty = true; break; }
                }
            }
            const unsigned long long dmask = __ballot(dirty);
            const int first_dirty = dmask ? __ffsll(dmask) - 1 : 64;
            ORBX_WAVE_SYNC();
            if (claims) s_claim[idx1] = 0xFFu;                         // (every claimer of a slot restores it: the slot is 0xFF again for the next round)
            ORBX_WAVE_SYNC();
            // 4. commit the lanes before the first dirty one
            const bool commit = pending && lane < first_dirty;
            if (commit && accept) {
                atomicMax(&assigned[idx1], qi);
                if (obs) atomicOr(&s_occ[idx1 >> 5], 1u << (idx1 & 31));
                if (LASTFRAME && check_ori) {                          // rot = angle(last) - angle(current), bin = round(rot / 30) (:2118-2125, the reference's 1 / HISTO_LENGTH factor)
                    float rot = __fsub_rn(last_angle[qi], cur_kps[idx1].angle);
                    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                    int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
                    if (bin == 30) bin = 0;
                    atomicAdd(&s_hist[bin & 31], 1);
                    events[qi] = idx1 | (bin << 16);
                }
            }
            nm += __popcll(__ballot(commit && accept));
            if (commit) pending = false;
            ORBX_WAVE_SYNC();
        }
    }
    if (LASTFRAME && check_ori) {
        ORBX_WAVE_SYNC();
        // ComputeThreeMaxima (src/ORBmatcher.cc:2335-2377) over the 30 bin sizes, every lane alike
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < 30; i++) {
            const int sz = s_hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        int taken_back = 0;
        for (int qi = lane; qi < M; qi += 64) {
            const int ev = events[qi];
            if (ev < 0) continue;
            const int bin = ev >> 16;
            if (bin != ind1 && bin != ind2 && bin != ind3) { assigned[ev & 0xFFFF] = -2; taken_back++; }
        }
        nm -= wave_sum(taken_back);
    }
    if (lane == 0) nmatches[b] = nm;
}

__global__ void __launch_bounds__(64) k_local_accept(int M, int cap, const int* __restrict__ n_per_frame, const int* __restrict__ q_start,
                                                     const int* __restrict__ q_count, const int2* __restrict__ entries, const uint8_t* __restrict__ occupied0,
                                                     const uint8_t* __restrict__ has_obs, float nnratio, int th_high, int* __restrict__ assigned,
                                                     int* __restrict__ nmatches) {
    local_accept_body<false>(M, cap, n_per_frame, q_start, q_count, entries, occupied0, has_obs, nnratio, th_high, assigned, nmatches, nullptr, nullptr, 0);
}
__global__ void __launch_bounds__(64) k_lastframe_accept(int M, int cap, const int* __restrict__ n_per_frame, const int* __restrict__ q_start,
                                                         const int* __restrict__ q_count, const int2* __restrict__ entries, const uint8_t* __restrict__ occupied0,
                                                         const uint8_t* __restrict__ has_obs, int th_high, int* __restrict__ assigned, int* __restrict__ nmatches,
                                                         const float* __restrict__ last_angle, const KeyPointRec* __restrict__ cur_kps, int check_ori) {
    local_accept_body<true>(M, cap, n_per_frame, q_start, q_count, entries, occupied0, has_obs, 0.0f, th_high, assigned, nmatches, last_angle, cur_kps, check_ori);
}

// Frame::ComputeStereoFromRGBD (src/Frame.cc:1361-1391) for B frames: mvDepth[i] = imDepth.at<float>(v, u) at the (distorted) keypoint, truncated
// to integers like cv::Mat::at(int, int) with float arguments, and mvuRight[i] = kpU.pt.x - mbf / d where d > 0, else both -1.  keys_un == NULL:
// no distortion (mvKeysUn = mvKeys).  depth image b at depth + b * image_stride (floats), rows `stride` floats apart.
__global__ void __launch_bounds__(256) k_stereo_from_depth(const KeyPointRec* __restrict__ kps, const KeyPointRec* __restrict__ kps_un, const int* __restrict__ n_per_frame,
                                                           int cap, const float* __restrict__ depth, int stride, size_t image_stride, int w, int h, float mbf,
                                                           float* __restrict__ u_right, float* __restrict__ depth_out, int* __restrict__ n_valid) {
    const size_t b = blockIdx.y;
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= cap) return;
    float ur = -1.0f, d_out = -1.0f;
    if (i < n_per_frame[b]) {
        const KeyPointRec kp = kps[b * (size_t)cap + i];
        const int u = (int)kp.x, v = (int)kp.y;
        if (u >= 0 && u < w && v >= 0 && v < h) {
            const float d = depth[b * image_stride + (size_t)v * stride + u];
            if (d > 0) {
                const float xu = kps_un ? kps_un[b * (size_t)cap + i].x : kp.x;
                d_out = d; ur = __fsub_rn(xu, __fdiv_rn(mbf, d));
                atomicAdd(&n_valid[b], 1);
            }
        }
    }
    u_right[b * (size_t)cap + i] = ur; depth_out[b * (size_t)cap + i] = d_out;
}

// One wave per work item (an unmatched feature idx1 of KF1 and the feature list of a neighbour KF2 in the same node; the arrays of
// all neighbours of a batch are concatenated, item.out_off = neighbour).  best2[item] = chosen (global) idx2 or -1.   src/ORBmatcher.cc:1117-1254
// The best candidate for feature idx1 of KF1 (record k1, descriptor a0..a3) among cnt2 features of KF2 listed in feat2 (one vocabulary
// node): src/ORBmatcher.cc:1117-1254.  Returns the chosen entry of feat2 or -1.  One wave; every lane gets the result.
template <bool KB8>
__device__ __forceinline__ int bow_search_core(int idx1, const KeyPointRec k1, bool stereo1, unsigned long long a0, unsigned long long a1,
                                               unsigned long long a2, unsigned long long a3, const BowParams& P,
                                               const KeyPointRec* __restrict__ kps2, const unsigned long long* __restrict__ desc2,
                                               const float* __restrict__ ur2, const uint8_t* __restrict__ has_mp2,
                                               const int* __restrict__ feat2, int cnt2) {
    const int lane = lane_id();
    // epipolar line of kp1 in image 2 (Pinhole::epipolarConstrain, src/CameraModels/Pinhole.cpp:186-217)
    const float la = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, P.F12[0]), __fmul_rn(k1.y, P.F12[3])), P.F12[6]);
    const float lb = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, P.F12[1]), __fmul_rn(k1.y, P.F12[4])), P.F12[7]);
    const float lc = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, P.F12[2]), __fmul_rn(k1.y, P.F12[5])), P.F12[8]);
    const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
    // Kannala-Brandt cameras: which camera of a rig the feature belongs to (:1136-1141), its ray (unprojected once per feature)
    const bool rig = KB8 && P.nleft1 >= 0 && P.nleft2 >= 0;                          // pKF1->mpCamera2 && pKF2->mpCamera2 (:1203)
    const int right1 = (KB8 && P.nleft1 >= 0 && idx1 >= P.nleft1) ? 1 : 0;
    KB8Cam c1; float ray1[3] = {0.f, 0.f, 1.f};
    if (KB8) {
#pragma unroll
        for (int i = 0; i < 8; i++) c1.p[i] = P.cam1[rig ? right1 : 0][i];
        kb8_unproject(c1, k1.x, k1.y, ray1);
    }
    unsigned best = 0xFFFFFFFFu;   // dist << 16 | (0xFFFF - position): the LAST candidate with the smallest distance wins (:1178 '>' test)
    for (int j = lane; j < cnt2; j += 64) {
        const int idx2 = feat2[j];
        if (has_mp2[idx2]) continue;
        const bool stereo2 = ur2[idx2] >= 0;
        if (P.only_stereo && !stereo2) continue;
        const unsigned long long* db = desc2 + 4 * (size_t)idx2;
        const int dist = __popcll(a0 ^ db[0]) + __popcll(a1 ^ db[1]) + __popcll(a2 ^ db[2]) + __popcll(a3 ^ db[3]);
        if (dist > P.th_low) continue;
        const KeyPointRec k2 = kps2[idx2];
        if (!stereo1 && !stereo2 && !(KB8 && P.nleft1 >= 0)) {                       // ... && !pKF1->mpCamera2 (:1189)
            const float dex = __fsub_rn(P.ep[0], k2.x), dey = __fsub_rn(P.ep[1], k2.y);
            if (__fadd_rn(__fmul_rn(dex, dex), __fmul_rn(dey, dey)) < __fmul_rn(100.f, P.scale2[k2.octave])) continue;
        }
        if (KB8) {
            if (!P.coarse) {
                const int right2 = (P.nleft2 >= 0 && idx2 >= P.nleft2) ? 1 : 0;
                const int sel = rig ? right1 * 2 + right2 : 0;
                KB8Cam c2; float ray2[3], p3D[3];
#pragma unroll
                for (int i = 0; i < 8; i++) c2.p[i] = P.cam2[rig ? right2 : 0][i];
                kb8_unproject(c2, k2.x, k2.y, ray2);
                const float z = kb8_triangulate_matches(c1, c2, ray1, ray2, k1.x, k1.y, k2.x, k2.y, P.R[sel], P.t[sel], P.sigma2_1[k1.octave], P.sigma2_2[k2.octave], p3D);
                if (!(z > 0.0001f)) continue;                                            // KannalaBrandt8::epipolarConstrain (:322-328)
            }
        } else if (!P.coarse) {
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, k2.x), __fmul_rn(lb, k2.y)), lc);
            if (den == 0) continue;
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            if (!((double)dsqr < 3.84 * (double)P.sigma2_2[k2.octave])) continue;
        }
        const unsigned key = ((unsigned)dist << 16) | (unsigned)(0xFFFF - j);
        best = key < best ? key : best;
    }
    best = wave_min_u32(best);
    return best == 0xFFFFFFFFu ? -1 : feat2[0xFFFF - (int)(best & 0xFFFF)];
}

// One wave per work item (an unmatched feature idx1 of KF1 and the feature list of a neighbour KF2 in the same node; the arrays of
// all neighbours of a batch are concatenated, item.out_off = neighbour).  best2[item] = chosen (global) idx2 or -1.
template <bool KB8>
__device__ __forceinline__ void bow_search_body(const BowItem* __restrict__ items, int nitems,
                                                    const KeyPointRec* __restrict__ kps1, const unsigned long long* __restrict__ desc1,
                                                    const float* __restrict__ ur1,
                                                    const KeyPointRec* __restrict__ kps2, const unsigned long long* __restrict__ desc2,
                                                    const float* __restrict__ ur2, const uint8_t* __restrict__ has_mp2,
                                                    const int* __restrict__ feat2, const BowParams* __restrict__ Ps, int* __restrict__ best2) {
    const int lane = lane_id();
    const int it = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (it >= nitems) return;
    const BowItem I = items[it];
    const BowParams& P = Ps[I.out_off];            // the neighbour key frame this item belongs to (wave-uniform)
    const unsigned long long* da = desc1 + 4 * (size_t)I.idx1;
    const int r = bow_search_core<KB8>(I.idx1, kps1[I.idx1], ur1[I.idx1] >= 0, da[0], da[1], da[2], da[3], P, kps2, desc2, ur2, has_mp2, feat2 + I.start2, I.cnt2);
    if (lane == 0) best2[it] = r;
}

// Position of `id` in the ascending list ids[0, n), or -1: two 64-way probes by the whole wave instead of a dependent binary search.
__device__ __forceinline__ int wave_find_node(const uint32_t* __restrict__ ids, int n, uint32_t id) {
    const int lane = lane_id();
    int lo = 0, len = n;
    while (len > 64) {
        const int stride = (len + 63) >> 6;
        const int pos = lo + lane * stride;
        const unsigned long long le = __ballot(pos < lo + len && ids[pos] <= id);
        if (le == 0ull) return -1;
        const int seg = 63 - __clzll((long long)le);          // last probe <= id
        const int nlo = lo + seg * stride;
        len = imin(stride, lo + len - nlo); lo = nlo;
    }
    const unsigned long long eq = __ballot(lane < len && ids[lo + lane] == id);
    return eq ? lo + (__ffsll((long long)eq) - 1) : -1;
}

// ORBmatcher::SearchForTriangulation over device-resident key frames: one wave per (feature idx1 of KF1, neighbour j), no work list - the
// wave finds the neighbour's feature list of idx1's vocabulary node itself.  best[j * N1 + idx1] = index in neighbour j or -1.
// grid (ceil(N1 / 4), n2).  flags: call-time map point flags, KF1's at offset 0, neighbour j's at nb[j].mp2_off.
template <bool KB8>
__device__ __forceinline__ void sft_resident_body(const ResidentKF& k1, const uint8_t* __restrict__ flags, const SftNeighbour* __restrict__ nb, int* __restrict__ best) {
    const int lane = lane_id();
    const int idx1 = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), j = (int)blockIdx.y;
    if (idx1 >= k1.N) return;
    int r = -1;
    const SftNeighbour& S = nb[j];
    const int a = k1.node_of_feat[idx1];
    const bool stereo1 = k1.ur[idx1] >= 0;
    if (a >= 0 && !flags[idx1] && !(S.P.only_stereo && !stereo1)) {
        const int b = wave_find_node(S.k2.node_id, S.k2.fv_nodes, k1.node_id[a]);
        if (b >= 0) {
            const int s2 = S.k2.fv_start[b], c2 = S.k2.fv_start[b + 1] - s2;
            const unsigned long long* da = k1.desc + 4 * (size_t)idx1;
            r = bow_search_core<KB8>(idx1, k1.kps[idx1], stereo1, da[0], da[1], da[2], da[3], S.P, S.k2.kps, S.k2.desc, S.k2.ur, flags + S.mp2_off,
                                     S.k2.fv_feat + s2, c2);
        }
    }
    if (lane == 0) best[(size_t)j * k1.N + idx1] = r;
}
__global__ void __launch_bounds__(256) k_sft_resident(ResidentKF k1, const uint8_t* __restrict__ flags, const SftNeighbour* __restrict__ nb,
                                                      int* __restrict__ best) {
    sft_resident_body<false>(k1, flags, nb, best);
}
// Kannala-Brandt cameras (its own kernel for the same reason as k_bow_search_kb8)
__global__ void __launch_bounds__(256) k_sft_resident_kb8(ResidentKF k1, const uint8_t* __restrict__ flags, const SftNeighbour* __restrict__ nb,
                                                          int* __restrict__ best) {
    sft_resident_body<true>(k1, flags, nb, best);
}

// ORBmatcher::SearchByBoW (src/ORBmatcher.cc:259-493 and :892-1043, non-fisheye) over device-resident key frames, the sequential accept
// loop included: a target taken by an earlier feature is skipped (:331, :948), and since a feature of K2 belongs to exactly one vocabulary
// node, that dependence never leaves a node - one wave per (node of K1, pair) walks the node's K1 features in order.
// A lane owns the candidates (positions in K2's list of the node) lane, lane + 64, ..; bit r of `taken` = position lane + 64 r is taken.
// m12[p * N1cap + idx1] = matched index in K2 (pre-set to -1).  Nodes with more than 2048 features in K2 set status bit 2.
// grid (ceil(max fv_nodes / 4), n)
__global__ void __launch_bounds__(256) k_bow_match_resident(const BowPairResident* __restrict__ pairs, const uint8_t* __restrict__ flags,
                                                            float nnratio, int th_low, int th_inclusive, int* __restrict__ m12, int N1cap,
                                                            int* __restrict__ status) {
    const int lane = lane_id();
    const int a = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), p = (int)blockIdx.y;
    const BowPairResident& B = pairs[p];
    if (a >= B.k1.fv_nodes || B.mp1_off < 0) return;
    const int b = wave_find_node(B.k2.node_id, B.k2_nodes_dev ? *B.k2_nodes_dev : B.k2.fv_nodes, B.k1.node_id[a]);
    if (b < 0) return;
    const int s2 = B.k2.fv_start[b], c2 = B.k2.fv_start[b + 1] - s2;
    if (c2 > 2048) { if (lane == 0) atomicOr(status, 4); return; }
    const uint8_t* mp1 = flags + B.mp1_off;
    const uint8_t* el2 = B.elig2_off >= 0 ? flags + B.elig2_off : nullptr;
    // this lane's candidates: index in K2 and eligibility, fetched once
    unsigned taken = 0;
    for (int k = B.k1.fv_start[a]; k < B.k1.fv_start[a + 1]; k++) {
        const int idx1 = B.k1.fv_feat[k];
        if (!mp1[idx1]) continue;                                                      // !pMP || pMP->isBad()
        const unsigned long long* da = B.k1.desc + 4 * (size_t)idx1;
        const unsigned long long a0 = da[0], a1 = da[1], a2 = da[2], a3 = da[3];
        unsigned k1key = (256u << 16) | 0xFFFFu, k2key = k1key;                        // this lane's two smallest (distance << 16 | position)
        for (int j = lane, r = 0; j < c2; j += 64, r++) {
            if ((taken >> r) & 1u) continue;
            const int idx2 = B.k2.fv_feat[s2 + j];
            if (el2 && !el2[idx2]) continue;
            const unsigned long long* db = B.k2.desc + 4 * (size_t)idx2;
            const unsigned d = (unsigned)(__popcll(a0 ^ db[0]) + __popcll(a1 ^ db[1]) + __popcll(a2 ^ db[2]) + __popcll(a3 ^ db[3]));
            const unsigned key = (d << 16) | (unsigned)j;
            if (key < k1key) { k2key = k1key; k1key = key; } else if (key < k2key) k2key = key;
        }
        // sequential rule: bestDist1 = smallest distance (first position that attains it), bestDist2 = second smallest of the multiset
        const unsigned m1 = wave_min_u32(k1key);
        const unsigned m2 = wave_min_u32(k1key == m1 ? k2key : k1key);
        const int bestDist1 = (int)(m1 >> 16), bestDist2 = (int)(m2 >> 16);
        const bool pass = th_inclusive ? bestDist1 <= th_low : bestDist1 < th_low;
        if (pass && (float)bestDist1 < __fmul_rn(nnratio, (float)bestDist2)) {
            const int j1 = (int)(m1 & 0xFFFFu);
            if (lane == (j1 & 63)) { taken |= 1u << (j1 >> 6); m12[(size_t)p * N1cap + idx1] = B.k2.fv_feat[s2 + j1]; }
        }
    }
}

// The rotation-consistency pruning behind SearchByBoW (src/ORBmatcher.cc:396-412 fill, :457-479 prune; ComputeThreeMaxima :2335-2377) on the device, one
// wave per pair: histogram of round((angle1 - angle2 [+ 360]) / 30) over the matches, entries outside the three strongest bins (the second / third
// only while they hold at least 10 % of the first) are reset to -1; nmatches[p] = matches left.  m12 as written by k_bow_match_resident.
__global__ void __launch_bounds__(64) k_bow_rotation_prune(const BowPairResident* __restrict__ pairs, int* __restrict__ m12, int N1cap, int check_ori,
                                                           int* __restrict__ nmatches) {
    __shared__ int s_hist[32];
    const int lane = lane_id(), p = (int)blockIdx.x;
    const BowPairResident& B = pairs[p];
    const int N1 = B.k1.N;
    int* m = m12 + (size_t)p * N1cap;
    if (lane < 32) s_hist[lane] = 0;
    ORBX_WAVE_SYNC();
    auto bin_of = [&](int i, int j) {
        float rot = __fsub_rn(B.k1.kps[i].angle, B.k2.kps[j].angle);
        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
        int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
        return bin == 30 ? 0 : bin;
    };
    int nm = 0;
    for (int i = lane; i < N1; i += 64) {
        const int j = m[i];
        if (j < 0) continue;
        nm++;
        if (check_ori) atomicAdd(&s_hist[bin_of(i, j) & 31], 1);
    }
    ORBX_WAVE_SYNC();
    if (check_ori) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < 30; i++) {
            const int sz = s_hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        for (int i = lane; i < N1; i += 64) {
            const int j = m[i];
            if (j < 0) continue;
            const int bin = bin_of(i, j);
            if (bin != ind1 && bin != ind2 && bin != ind3) { m[i] = -1; nm--; }
        }
    }
    nm = wave_sum(nm);
    if (lane == 0) nmatches[p] = nm;
}

__global__ void __launch_bounds__(256) k_bow_search(const BowItem* __restrict__ items, int nitems, const KeyPointRec* __restrict__ kps1,
                                                    const unsigned long long* __restrict__ desc1, const float* __restrict__ ur1,
                                                    const KeyPointRec* __restrict__ kps2, const unsigned long long* __restrict__ desc2,
                                                    const float* __restrict__ ur2, const uint8_t* __restrict__ has_mp2,
                                                    const int* __restrict__ feat2, const BowParams* __restrict__ Ps, int* __restrict__ best2) {
    bow_search_body<false>(items, nitems, kps1, desc1, ur1, kps2, desc2, ur2, has_mp2, feat2, Ps, best2);
}
// the same with Kannala-Brandt cameras (the triangulation test needs ~200 registers: its own kernel keeps the pinhole one slim)
__global__ void __launch_bounds__(256) k_bow_search_kb8(const BowItem* __restrict__ items, int nitems, const KeyPointRec* __restrict__ kps1,
                                                        const unsigned long long* __restrict__ desc1, const float* __restrict__ ur1,
                                                        const KeyPointRec* __restrict__ kps2, const unsigned long long* __restrict__ desc2,
                                                        const float* __restrict__ ur2, const uint8_t* __restrict__ has_mp2,
                                                        const int* __restrict__ feat2, const BowParams* __restrict__ Ps, int* __restrict__ best2) {
    bow_search_body<true>(items, nitems, kps1, desc1, ur1, kps2, desc2, ur2, has_mp2, feat2, Ps, best2);
}

// All distances of one unmatched feature idx1 to the features of the other key frame / frame in the same vocabulary node, in the
// node's order, for the sequential accept loops of ORBmatcher::SearchByBoW (src/ORBmatcher.cc:259-493 and :892-1043), which skip
// targets taken by earlier features.  out[item.out_off + j] = Hamming distance, or -1 when the j-th target is not eligible.
__global__ void __launch_bounds__(256) k_bow_dists(const BowItem* __restrict__ items, int nitems,
                                                   const unsigned long long* __restrict__ desc1, const unsigned long long* __restrict__ desc2,
                                                   const uint8_t* __restrict__ eligible2, const int* __restrict__ feat2, int* __restrict__ out) {
    const int lane = lane_id();
    const int it = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (it >= nitems) return;
    const BowItem I = items[it];
    const unsigned long long* da = desc1 + 4 * (size_t)I.idx1;
    const unsigned long long a0 = da[0], a1 = da[1], a2 = da[2], a3 = da[3];
    for (int j = lane; j < I.cnt2; j += 64) {
        const int idx2 = feat2[I.start2 + j];
        int d = -1;
        if (eligible2[idx2]) {
            const unsigned long long* db = desc2 + 4 * (size_t)idx2;
            d = __popcll(a0 ^ db[0]) + __popcll(a1 ^ db[1]) + __popcll(a2 ^ db[2]) + __popcll(a3 ^ db[3]);
        }
        out[I.out_off + j] = d;
    }
}

}  // namespace orbx
