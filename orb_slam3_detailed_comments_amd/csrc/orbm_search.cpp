// orbm_search.cpp — host side of the guided searches (include/orbx.h "Guided searches"): uploads the SoA views,
// launches k_grid_build / k_area_search / k_bow_search, and replays the reference's *order-dependent* acceptance loops
// over the device-computed candidate lists:
//   SearchByProjection(Frame, MapPoints)   src/ORBmatcher.cc:45-239
//   SearchByProjection(Frame, Frame)       src/ORBmatcher.cc:1950-2184 (+ ComputeThreeMaxima :2335-2377)
//   SearchForTriangulation                 src/ORBmatcher.cc:1045-1323
// All Hamming distances and all window / level / gate / epipolar tests run on the GPU; the replay below only compares
// precomputed integers in the reference's sequence (an assignment for one map point changes the candidate set of the next).
#include "orbx_internal.h"

using namespace orbx;

namespace {

const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;   // src/ORBmatcher.cc:35-37
const int kGridColsHost = 64, kGridRowsHost = 48;          // FRAME_GRID_COLS / FRAME_GRID_ROWS, include/Frame.h:44-45

struct DeviceFrame {
    const KeyPointRec* kps; const unsigned long long* desc; const float* ur; const int* cell_start; const int* cell_items;
    GridParams g;
};

// scratch slots in orbx_extractor::d_sr / d_si
enum { SR_KPS = 0, SR_DESC, SR_UR, SR_QUERY, SR_QDESC, SR_ENTRIES, SR_KPS2, SR_DESC2, SR_UR2, SR_HASMP2, SR_ITEMS, SR_SPARE };
enum { SI_CELLOF = 0, SI_CELLSTART, SI_CELLITEMS, SI_QSTART, SI_QCOUNT, SI_COUNTER, SI_FEAT2, SI_BEST };

int upload(orbx_extractor* h, int slot, const void* src, size_t bytes) {
    if (h->d_sr[slot].ensure(bytes + 16)) return -1;
    return rt::copy_h2d(h->d_sr[slot].p, src, bytes, h->s0);
}

// Per call the window searches move: frame (keypoints | descriptors | uRight) and queries (AreaQuery | descriptors) up, the candidate
// lists down.  Each direction is ONE async copy between a pinned staging buffer and one device buffer (a call is latency-bound: with a
// dozen small pageable copies the driver calls cost more than the kernels).
inline size_t al16(size_t v) { return (v + 15) & ~(size_t)15; }

// several host arrays -> one pinned staging buffer -> one async copy into d_sr[SR_KPS]; add() returns the device address
struct Packer {
    orbx_extractor* h; std::vector<std::pair<const void*, size_t>> parts; std::vector<size_t> offs; size_t total = 0;
    explicit Packer(orbx_extractor* hh) : h(hh) {}
    size_t add(const void* src, size_t bytes) { offs.push_back(total); parts.emplace_back(src, bytes); total += al16(bytes ? bytes : 1); return offs.size() - 1; }
    int flush() {
        if (h->h_packA.ensure(total + 16) || h->d_sr[SR_KPS].ensure(total + 16)) return -1;
        for (size_t i = 0; i < parts.size(); i++) if (parts[i].second) memcpy(h->h_packA.p + offs[i], parts[i].first, parts[i].second);
        return rt::copy_h2d(h->d_sr[SR_KPS].p, h->h_packA.p, total, h->s0);
    }
    template <typename T> const T* dev(size_t part) const { return (const T*)(h->d_sr[SR_KPS].p + offs[part]); }
};

int upload_frame(orbx_extractor* h, const OrbmFrameView* F, DeviceFrame* D) {
    if (!F || F->N < 0 || (F->N > 0 && (!F->keys_un || !F->desc))) return fail(ORBX_E_ARG, "bad frame view");
    if (F->N >= 65535) return fail(ORBX_E_ARG, "too many keypoints");
    const int N = F->N, N1 = N > 0 ? N : 1;
    const size_t okp = 0, odesc = al16(sizeof(KeyPointRec) * (size_t)N1), our = odesc + al16(32 * (size_t)N1), total = our + al16(sizeof(float) * (size_t)N1);
    int e = h->h_packA.ensure(total + 16) | h->d_sr[SR_KPS].ensure(total + 16);
    e |= h->d_si[SI_CELLOF].ensure(N + 1) | h->d_si[SI_CELLSTART].ensure(64 * 48 + 2) | h->d_si[SI_CELLITEMS].ensure(N + 1) | h->d_si[SI_COUNTER].ensure(4);
    if (e) return fail(ORBX_E_DEVICE, "upload/allocation failed");
    uint8_t* hp = h->h_packA.p;
    if (N > 0) { memcpy(hp + okp, F->keys_un, sizeof(KeyPointRec) * (size_t)N); memcpy(hp + odesc, F->desc, 32 * (size_t)N); }
    float* ur = (float*)(hp + our);
    if (F->u_right) memcpy(ur, F->u_right, sizeof(float) * (size_t)N); else for (int i = 0; i < N1; i++) ur[i] = -1.0f;
    if (rt::copy_h2d(h->d_sr[SR_KPS].p, hp, total, h->s0)) return fail(ORBX_E_DEVICE, "upload failed");
    const uint8_t* dp = h->d_sr[SR_KPS].p;
    D->kps = (const KeyPointRec*)(dp + okp); D->desc = (const unsigned long long*)(dp + odesc); D->ur = (const float*)(dp + our);
    memset(&D->g, 0, sizeof D->g);
    D->g.min_x = F->min_x; D->g.min_y = F->min_y; D->g.gw_inv = F->grid_w_inv; D->g.gh_inv = F->grid_h_inv;
    const dim3 one(1, 1, 1), blkg(kGridThreads, 1, 1);
    ORBX_LAUNCH(k_grid_build, one, blkg, 0, h->s0, D->kps, N, D->g, h->d_si[SI_CELLOF].p, h->d_si[SI_CELLSTART].p, h->d_si[SI_CELLITEMS].p, (const int*)nullptr, 0);
    D->cell_start = h->d_si[SI_CELLSTART].p; D->cell_items = h->d_si[SI_CELLITEMS].p;
    return ORBX_OK;
}

struct Csr { std::vector<int> start, count; std::vector<int> ent; };   // ent: 2 ints per candidate {idx, dist | octave << 16}
void fill_frustum_params(const OrbmFrustumView* V, float cos_limit, float th, int far_points, float th_far, FrustumParams* Fp) {
    memset(Fp, 0, sizeof *Fp);
    memcpy(Fp->Rcw, V->Rcw, sizeof Fp->Rcw); memcpy(Fp->tcw, V->tcw, sizeof Fp->tcw); memcpy(Fp->Ow, V->Ow, sizeof Fp->Ow); memcpy(Fp->qcw, V->qcw, sizeof Fp->qcw);
    memcpy(Fp->cam, V->cam, sizeof Fp->cam); Fp->kb8 = V->camera_type == 1;
    Fp->min_x = V->min_x; Fp->max_x = V->max_x; Fp->min_y = V->min_y; Fp->max_y = V->max_y; Fp->mbf = V->mbf; Fp->log_scale_factor = V->log_scale_factor; Fp->nlevels = V->nlevels;
    for (int l = 0; l < V->nlevels; l++) Fp->scale_factors[l] = V->scale_factors[l];
    Fp->cos_limit = cos_limit; Fp->th = th; Fp->th_far = th_far; Fp->far_points = far_points;
}

// runs k_area_search for Q queries.  Results come back in one copy: [total, -, -, -][start Q][count Q][entries]; the number of
// entries fetched with the header is a guess from the previous call, a second copy follows only if it was too small, and the pool is
// grown and the search repeated if the pool itself overflowed.
int run_area_search_dev(orbx_extractor* h, const DeviceFrame& D, int Q, const AreaQuery* dq, const unsigned long long* dqd, Csr* out, const void* extra_src = nullptr,
                        size_t extra_bytes = 0, const uint8_t** extra_host = nullptr);
int run_area_search(orbx_extractor* h, const DeviceFrame& D, const std::vector<AreaQuery>& qs, const uint8_t* qdesc, Csr* out) {
    const int Q = (int)qs.size();
    out->start.assign(Q, 0); out->count.assign(Q, 0); out->ent.clear();
    if (Q == 0) return ORBX_OK;
    const size_t oq = 0, oqd = al16(sizeof(AreaQuery) * (size_t)Q), qtotal = oqd + al16(32 * (size_t)Q);
    if (h->h_packB.ensure(qtotal + 16) || h->d_sr[SR_QUERY].ensure(qtotal + 16)) return fail(ORBX_E_DEVICE, "upload/allocation failed");
    memcpy(h->h_packB.p + oq, qs.data(), sizeof(AreaQuery) * (size_t)Q); memcpy(h->h_packB.p + oqd, qdesc, 32 * (size_t)Q);
    if (rt::copy_h2d(h->d_sr[SR_QUERY].p, h->h_packB.p, qtotal, h->s0)) return fail(ORBX_E_DEVICE, "upload failed");
    return run_area_search_dev(h, D, Q, (const AreaQuery*)(h->d_sr[SR_QUERY].p + oq), (const unsigned long long*)(h->d_sr[SR_QUERY].p + oqd), out);
}
// the same with the queries and their descriptors already on the device.  extra_src / extra_bytes: another device block to bring back with the
// same synchronisation (the tracking fields of SearchLocalPoints); *extra_host points at its pinned copy afterwards.
int run_area_search_dev(orbx_extractor* h, const DeviceFrame& D, int Q, const AreaQuery* dq, const unsigned long long* dqd, Csr* out, const void* extra_src,
                        size_t extra_bytes, const uint8_t** extra_host) {
    out->start.assign(Q, 0); out->count.assign(Q, 0); out->ent.clear();
    if (Q == 0) return ORBX_OK;
    const size_t hdr = 16 + 8 * (size_t)Q;                          // bytes in front of the entries
    size_t pool = std::max<size_t>(h->area_pool, (size_t)Q * 48 + 1024);
    for (int attempt = 0; attempt < 2; attempt++) {
        if (h->d_sr[SR_ENTRIES].ensure(hdr + pool * 8 + 16)) return fail(ORBX_E_DEVICE, "entry pool allocation failed");
        h->area_pool = pool;
        uint8_t* dout = h->d_sr[SR_ENTRIES].p;
        int* d_counter = (int*)dout; int* d_start = (int*)(dout + 16); int* d_count = d_start + Q; int2* d_ent = (int2*)(dout + hdr);
        rt::memset_async(d_counter, 0, 16, h->s0);
        dim3 grid((Q + kAreaWaves - 1) / kAreaWaves, 1, 1), blk(64 * kAreaWaves, 1, 1);
        ORBX_LAUNCH(k_area_search, grid, blk, 0, h->s0, dq, dqd, Q, D.kps, D.ur, D.desc, D.g, D.cell_start, D.cell_items, 1, d_counter, (int)pool,
                    d_start, d_count, d_ent, 0);
        const size_t guess = std::min(pool, std::max<size_t>(h->area_last_total + h->area_last_total / 4 + 256, 1024));
        const size_t oextra = al16(hdr + pool * 8 + 16);
        if (h->h_out.ensure(oextra + extra_bytes + 16)) return fail(ORBX_E_DEVICE, "pinned allocation failed");
        rt::copy_d2h(h->h_out.p, dout, hdr + guess * 8, h->s0);
        if (extra_bytes) { rt::copy_d2h(h->h_out.p + oextra, extra_src, extra_bytes, h->s0); if (extra_host) *extra_host = h->h_out.p + oextra; }
        if (rt::stream_sync(h->s0) || rt::check_launch()) return fail(ORBX_E_DEVICE, "area search failed: %s", rt::last_error());
        const int total = *(const int*)h->h_out.p;
        if ((size_t)total <= pool) {
            if ((size_t)total > guess) {
                rt::copy_d2h(h->h_out.p + hdr + guess * 8, dout + hdr + guess * 8, ((size_t)total - guess) * 8, h->s0);
                if (rt::stream_sync(h->s0)) return fail(ORBX_E_DEVICE, "area search download failed: %s", rt::last_error());
            }
            h->area_last_total = (size_t)total;
            memcpy(out->start.data(), h->h_out.p + 16, sizeof(int) * (size_t)Q);
            memcpy(out->count.data(), h->h_out.p + 16 + sizeof(int) * (size_t)Q, sizeof(int) * (size_t)Q);
            out->ent.resize(2 * (size_t)total + 2);
            if (total > 0) memcpy(out->ent.data(), h->h_out.p + hdr, 8 * (size_t)total);
            return ORBX_OK;
        }
        pool = (size_t)total + 1024;
    }
    return fail(ORBX_E_INTERNAL, "entry pool overflow after retry");
}

// ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:2335-2377
void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

int rot_bin(float a1, float a2) {    // :2118-2123; the 1/HISTO_LENGTH factor is the reference's (kept on purpose)
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

}  // namespace

extern "C" {

int orbm_get_features_in_area(orbx_extractor* h, const OrbmFrameView* F, float x, float y, float r, int min_level, int max_level, int* indices, int cap) {
    if (!h || !indices) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    DeviceFrame D;
    int rc = upload_frame(h, F, &D); if (rc) return rc;
    std::vector<AreaQuery> qs(1);
    qs[0].x = x; qs[0].y = y; qs[0].r = r; qs[0].ur = 0; qs[0].min_level = min_level; qs[0].max_level = max_level; qs[0].active = 1; qs[0].gate = 0;
    uint8_t zero[32] = {0};
    Csr c;
    rc = run_area_search(h, D, qs, zero, &c); if (rc) return rc;
    for (int k = 0; k < c.count[0] && k < cap; k++) indices[k] = c.ent[2 * (size_t)(c.start[0] + k)];
    return c.count[0];
}

int orbm_area_search_batch(orbx_extractor* h, const OrbmFrameView* F, const OrbmAreaQuery* queries, const uint8_t* query_desc, int Q,
                           int* start, int* count, int* idx, int* dist, int* level, int cap) {
    if (!h || !F || (Q > 0 && (!queries || !query_desc || !start || !count))) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    DeviceFrame D;
    int rc = upload_frame(h, F, &D); if (rc) return rc;
    std::vector<AreaQuery> qs(Q);
    for (int i = 0; i < Q; i++) {
        AreaQuery& q = qs[i]; memset(&q, 0, sizeof q);
        q.x = queries[i].x; q.y = queries[i].y; q.r = queries[i].r; q.min_level = queries[i].min_level; q.max_level = queries[i].max_level;
        q.active = 1; q.gate = 0;
    }
    Csr c;
    rc = run_area_search(h, D, qs, query_desc, &c); if (rc) return rc;
    int total = 0;
    for (int i = 0; i < Q; i++) { start[i] = c.start[i]; count[i] = c.count[i]; total = std::max(total, c.start[i] + c.count[i]); }
    if (total <= cap && idx && dist && level)
        for (int k = 0; k < total; k++) { idx[k] = c.ent[2 * (size_t)k]; dist[k] = c.ent[2 * (size_t)k + 1] & 0xFFFF; level[k] = c.ent[2 * (size_t)k + 1] >> 16; }
    return total;
}

int orbm_search_by_projection_mappoints(orbx_extractor* h, const OrbmFrameView* F, const OrbmMapPointView* P, float th, int far_points,
                                        float th_far, float nnratio, int* assigned, int* nmatches_out) {
    if (!h || !F || !P || !assigned) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    DeviceFrame D;
    int rc = upload_frame(h, F, &D); if (rc) return rc;
    const int M = P->M, N = F->N;
    const bool bFactor = th != 1.0;
    std::vector<AreaQuery> qs(M);
    for (int i = 0; i < M; i++) {
        AreaQuery& q = qs[i]; memset(&q, 0, sizeof q);
        // :53-60 skip tests; (the right-camera branch :170-236 needs Nleft != -1 and is not part of this path)
        if (!P->in_view[i]) continue;
        if (far_points && P->track_depth[i] > th_far) continue;
        if (P->is_bad[i]) continue;
        const int lvl = P->scale_level[i];
        if (lvl < 0 || lvl >= F->nlevels) continue;
        float r = P->view_cos[i] > 0.998 ? 2.5f : 4.0f;                 // RadiusByViewingCos, :242-249
        if (bFactor) r *= th;
        q.x = P->proj_x[i]; q.y = P->proj_y[i]; q.r = r * F->scale_factors[lvl]; q.ur = P->proj_xr[i];
        q.min_level = lvl - 1; q.max_level = lvl; q.active = 1; q.gate = 1;
    }
    Csr c;
    rc = run_area_search(h, D, qs, P->desc, &c); if (rc) return rc;
    // ---- sequential replay of :62-166 ----
    std::vector<uint8_t> occ(N > 0 ? N : 1, 0);
    if (F->occupied) memcpy(occ.data(), F->occupied, N);
    for (int i = 0; i < N; i++) assigned[i] = -1;
    int nmatches = 0;
    for (int i = 0; i < M; i++) {
        if (!qs[i].active || c.count[i] == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < c.count[i]; k++) {
            const int idx = c.ent[2 * (size_t)(c.start[i] + k)], dl = c.ent[2 * (size_t)(c.start[i] + k) + 1];
            if (occ[idx]) continue;
            const int dist = dl & 0xFFFF, level = dl >> 16;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = level; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = level; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                assigned[bestIdx] = i;
                occ[bestIdx] = P->has_obs ? P->has_obs[i] : 1;
                nmatches++;
            }
        }
    }
    if (nmatches_out) *nmatches_out = nmatches;
    return ORBX_OK;
}

// ---- C1 on the device: Frame::isInFrustum + MapPoint::PredictScale (k_frustum), alone or in front of SearchByProjection(Frame, MapPoints) ----
namespace {
struct FrustumDev { uint8_t* in_view; float* track; int* level; AreaQuery* queries; const unsigned long long* qdesc; };
int enqueue_frustum(orbx_extractor* h, const OrbmFrustumView* V, const OrbmWorldPointView* P, float cos_limit, bool with_queries, float th, int far_points, float th_far,
                    FrustumDev* out, const FrustumParams* second = nullptr, FrustumDev* out2 = nullptr) {
    if (!V || !P || P->M < 0 || (P->M > 0 && (!P->pos || !P->normal || !P->min_distance || !P->max_distance))) return fail(ORBX_E_ARG, "bad frustum arguments");
    if (V->nlevels < 1 || V->nlevels > kMaxLevels || !V->scale_factors) return fail(ORBX_E_ARG, "bad scale levels");
    const int M = P->M; const size_t M1 = M > 0 ? M : 1;
    // inputs -> d_sr[SR_QUERY]: [queries][descriptors][pos][normal][min][max][bad]; outputs -> d_sr[SR_SPARE]: [track 5 M][level M][in_view M]
    const size_t oq = 0, od = oq + al16(sizeof(AreaQuery) * M1), op = od + al16(32 * M1), on = op + al16(12 * M1), omn = on + al16(12 * M1), omx = omn + al16(4 * M1),
                 ob = omx + al16(4 * M1), total = ob + al16(M1);
    const size_t ot = 0, ol = ot + al16(20 * M1), ov = ol + al16(4 * M1), ototal = ov + al16(M1);
    if (h->h_packB.ensure(total + 16) || h->d_sr[SR_QUERY].ensure(total + 16) || h->d_sr[SR_SPARE].ensure(2 * ototal + 16)) return fail(ORBX_E_DEVICE, "upload/allocation failed");
    uint8_t* hp = h->h_packB.p;
    if (M > 0) {
        if (P->desc) memcpy(hp + od, P->desc, 32 * (size_t)M); else if (with_queries) return fail(ORBX_E_ARG, "map point descriptors missing");
        memcpy(hp + op, P->pos, 12 * (size_t)M); memcpy(hp + on, P->normal, 12 * (size_t)M);
        memcpy(hp + omn, P->min_distance, 4 * (size_t)M); memcpy(hp + omx, P->max_distance, 4 * (size_t)M);
        if (P->is_bad) memcpy(hp + ob, P->is_bad, M); else memset(hp + ob, 0, M);
    }
    if (rt::copy_h2d(h->d_sr[SR_QUERY].p + od, hp + od, total - od, h->s0)) return fail(ORBX_E_DEVICE, "upload failed");
    FrustumParams F; memset(&F, 0, sizeof F);
    memcpy(F.Rcw, V->Rcw, sizeof F.Rcw); memcpy(F.tcw, V->tcw, sizeof F.tcw); memcpy(F.Ow, V->Ow, sizeof F.Ow);
    memcpy(F.cam, V->cam, sizeof F.cam); F.kb8 = V->camera_type == 1;
    F.min_x = V->min_x; F.max_x = V->max_x; F.min_y = V->min_y; F.max_y = V->max_y; F.mbf = V->mbf; F.log_scale_factor = V->log_scale_factor; F.nlevels = V->nlevels;
    for (int l = 0; l < V->nlevels; l++) F.scale_factors[l] = V->scale_factors[l];
    F.cos_limit = cos_limit; F.th = th; F.th_far = th_far; F.far_points = far_points;
    F.rig_mode = second ? 1 : 0;
    uint8_t* di = h->d_sr[SR_QUERY].p; uint8_t* dout = h->d_sr[SR_SPARE].p;
    out->in_view = dout + ov; out->track = (float*)(dout + ot); out->level = (int*)(dout + ol);
    out->queries = with_queries ? (AreaQuery*)(di + oq) : nullptr; out->qdesc = (const unsigned long long*)(di + od);
    if (M > 0) {
        dim3 grid((M + 255) / 256, 1, 1), blk(256, 1, 1);
        ORBX_LAUNCH(k_frustum, grid, blk, 0, h->s0, F, M, (const float*)(di + op), (const float*)(di + on), (const float*)(di + omn), (const float*)(di + omx),
                    (const uint8_t*)(di + ob), out->in_view, out->track, out->level, out->queries, (int*)nullptr, (const FrustumParams*)nullptr);
    }
    if (second && out2) {                                           // the second camera of a rig: the same points, its own block behind the first one
        F.rig_mode = 1;                                             // (both cameras of a rig store nothing for a rejected point; the caller passed V with rig semantics)
        out2->in_view = dout + ototal + ov; out2->track = (float*)(dout + ototal + ot); out2->level = (int*)(dout + ototal + ol); out2->queries = nullptr; out2->qdesc = out->qdesc;
        if (M > 0) {
            dim3 grid((M + 255) / 256, 1, 1), blk(256, 1, 1);
            ORBX_LAUNCH(k_frustum, grid, blk, 0, h->s0, *second, M, (const float*)(di + op), (const float*)(di + on), (const float*)(di + omn), (const float*)(di + omx),
                        (const uint8_t*)(di + ob), out2->in_view, out2->track, out2->level, (AreaQuery*)nullptr, (int*)nullptr, (const FrustumParams*)nullptr);
        }
    }
    return ORBX_OK;
}
// hands the tracking fields (host copy of the frustum kernel's output block: [track 5 M][level M][in_view M]) to the caller's arrays
void scatter_track(const uint8_t* blk, int M, const OrbmTrackOut* out) {
    if (!out || M <= 0) return;
    const size_t M1 = M, ol = al16(20 * M1), ov = ol + al16(4 * M1);
    const float* tr = (const float*)blk;
    if (out->in_view) memcpy(out->in_view, blk + ov, M1);
    if (out->proj_x) memcpy(out->proj_x, tr, 4 * M1);
    if (out->proj_y) memcpy(out->proj_y, tr + M1, 4 * M1);
    if (out->proj_xr) memcpy(out->proj_xr, tr + 2 * M1, 4 * M1);
    if (out->depth) memcpy(out->depth, tr + 3 * M1, 4 * M1);
    if (out->view_cos) memcpy(out->view_cos, tr + 4 * M1, 4 * M1);
    if (out->scale_level) memcpy(out->scale_level, blk + ol, 4 * M1);
}
size_t track_block_bytes(int M) { const size_t M1 = M > 0 ? M : 1; return al16(20 * M1) + al16(4 * M1) + al16(M1); }
// brings the block back (one copy into pinned memory) and scatters it
int fetch_frustum(orbx_extractor* h, int M, const FrustumDev& D, const OrbmTrackOut* out) {
    if (M <= 0) return ORBX_OK;
    const size_t n = track_block_bytes(M);
    if (h->h_out.ensure(n + 16)) return fail(ORBX_E_DEVICE, "pinned allocation failed");
    if (rt::copy_d2h(h->h_out.p, D.track, n, h->s0) || rt::stream_sync(h->s0) || rt::check_launch()) return fail(ORBX_E_DEVICE, "frustum kernel failed: %s", rt::last_error());
    scatter_track(h->h_out.p, M, out);
    return ORBX_OK;
}
}  // namespace

int orbm_is_in_frustum(orbx_extractor* h, const OrbmFrustumView* V, const OrbmWorldPointView* P, float cos_limit, const OrbmTrackOut* out) {
    if (!h || !out) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    FrustumDev D;
    int rc = enqueue_frustum(h, V, P, cos_limit, false, 1.0f, 0, 0.0f, &D); if (rc) return rc;
    return fetch_frustum(h, P->M, D, out);
}

// ---- Frame::isInFrustum with two cameras (Nleft != -1, src/Frame.cc:754-766): isInFrustumChecks (:1592-1650) once per camera ----
namespace {
// a coefficient of an Eigen (>= 3.3) 3x3 product: a0 b0 + (a1 b1 + a2 b2) (sophus_action.h)
inline float dot3(const float* a, int sa, const float* b, int sb) { return a[0] * b[0] + (a[sa] * b[sb] + a[2 * sa] * b[2 * sb]); }
// camera 2 of the rig as isInFrustumChecks(pMP, cos, bRight = true) sets it up: mR = Rrl * mRcw, mt = Rrl * mtcw + trl, twc = mRwc * tlr + mOw
void right_camera_params(const OrbmFrustumRigView* V, float cos_limit, FrustumParams* Fp) {
    OrbmFrustumView R = V->left;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.Rcw[3 * i + j] = dot3(V->Rrl + 3 * i, 1, V->left.Rcw + j, 3);
    for (int i = 0; i < 3; i++) R.tcw[i] = dot3(V->Rrl + 3 * i, 1, V->left.tcw, 1) + V->trl[i];
    for (int i = 0; i < 3; i++) R.Ow[i] = dot3(V->Rwc + 3 * i, 1, V->tlr, 1) + V->left.Ow[i];
    R.camera_type = V->camera2_type; memcpy(R.cam, V->cam2, sizeof R.cam);
    fill_frustum_params(&R, cos_limit, 1.0f, 0, 0.0f, Fp);
    Fp->rig_mode = 1;
}
void scatter_track_right(const uint8_t* blk, int M, const OrbmTrackOutRight* out) {
    if (!out || M <= 0) return;
    const size_t M1 = M, ol = al16(20 * M1), ov = ol + al16(4 * M1);
    const float* tr = (const float*)blk;
    if (out->in_view_r) memcpy(out->in_view_r, blk + ov, M1);
    if (out->proj_xr) memcpy(out->proj_xr, tr, 4 * M1);
    if (out->proj_yr) memcpy(out->proj_yr, tr + M1, 4 * M1);
    if (out->depth_r) memcpy(out->depth_r, tr + 3 * M1, 4 * M1);
    if (out->view_cos_r) memcpy(out->view_cos_r, tr + 4 * M1, 4 * M1);
    if (out->scale_level_r) memcpy(out->scale_level_r, blk + ol, 4 * M1);
}
}  // namespace

int orbm_is_in_frustum_rig(orbx_extractor* h, const OrbmFrustumRigView* V, const OrbmWorldPointView* P, float cos_limit, const OrbmTrackOut* left, const OrbmTrackOutRight* right) {
    if (!h || !V || !P || !left || !right) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    FrustumParams Fr; right_camera_params(V, cos_limit, &Fr);
    FrustumDev D, D2;
    int rc = enqueue_frustum(h, &V->left, P, cos_limit, false, 1.0f, 0, 0.0f, &D, &Fr, &D2); if (rc) return rc;
    const int M = P->M;
    if (M <= 0) return ORBX_OK;
    const size_t n = track_block_bytes(M);
    if (h->h_out.ensure(2 * n + 16)) return fail(ORBX_E_DEVICE, "pinned allocation failed");
    if (rt::copy_d2h(h->h_out.p, D.track, 2 * n, h->s0) || rt::stream_sync(h->s0) || rt::check_launch()) return fail(ORBX_E_DEVICE, "frustum kernel failed: %s", rt::last_error());
    scatter_track(h->h_out.p, M, left);
    scatter_track_right(h->h_out.p + n, M, right);
    return ORBX_OK;
}

// ---- map points that stay on the device (the local map changes slowly; Tracking::SearchLocalPoints visits it every frame) ----
struct orbm_points {
    int device = 0, M = 0;
    uint8_t* dmem = nullptr;
    const unsigned long long* desc = nullptr; const float *pos = nullptr, *normal = nullptr, *min_d = nullptr, *max_d = nullptr;
};

int orbm_points_create(orbx_extractor* h, const OrbmWorldPointView* P, orbm_points** out) {
    if (!h || !P || !out || P->M < 0 || (P->M > 0 && (!P->pos || !P->normal || !P->min_distance || !P->max_distance || !P->desc))) return fail(ORBX_E_ARG, "bad map point view");
    rt::set_device(h->device);
    const size_t M1 = P->M > 0 ? P->M : 1;
    const size_t od = 0, op = od + al16(32 * M1), on = op + al16(12 * M1), omn = on + al16(12 * M1), omx = omn + al16(4 * M1), total = omx + al16(4 * M1);
    std::vector<uint8_t> st(total, 0);
    if (P->M > 0) {
        memcpy(&st[od], P->desc, 32 * (size_t)P->M); memcpy(&st[op], P->pos, 12 * (size_t)P->M); memcpy(&st[on], P->normal, 12 * (size_t)P->M);
        memcpy(&st[omn], P->min_distance, 4 * (size_t)P->M); memcpy(&st[omx], P->max_distance, 4 * (size_t)P->M);
    }
    orbm_points* r = new orbm_points();
    r->device = h->device; r->M = P->M; r->dmem = (uint8_t*)rt::dmalloc(total);
    if (!r->dmem || rt::copy_h2d(r->dmem, st.data(), total, h->s0) || rt::stream_sync(h->s0)) { rt::dfree(r->dmem); delete r; return fail(ORBX_E_DEVICE, "map point upload failed"); }
    r->desc = (const unsigned long long*)(r->dmem + od); r->pos = (const float*)(r->dmem + op); r->normal = (const float*)(r->dmem + on);
    r->min_d = (const float*)(r->dmem + omn); r->max_d = (const float*)(r->dmem + omx);
    *out = r;
    return ORBX_OK;
}

void orbm_points_destroy(orbm_points* p) {
    if (!p) return;
    rt::set_device(p->device);
    rt::dfree(p->dmem);
    delete p;
}

namespace {
// Tracking::SearchLocalPoints' device part (src/Tracking.cc:4009-4067): Frame::isInFrustum for every candidate map point, then
// ORBmatcher::SearchByProjection(Frame, MapPoints, th, bFarPoints, thFarPoints) on the points in view.  One upload (the frame, the call-time
// flags and - unless the points are resident - the points), three launches (grid, frustum + window queries, window search), one download
// (tracking fields, candidate lists), then the ordered replay of src/ORBmatcher.cc:62-166 on the host.
int search_local_points_core(orbx_extractor* h, const OrbmFrameView* F, const OrbmFrustumView* V, int M, const OrbmWorldPointView* P, const orbm_points* R,
                             const uint8_t* is_bad, const uint8_t* has_obs, float cos_limit, float th, int far_points, float th_far, float nnratio,
                             const OrbmTrackOut* out, int* assigned, int* nmatches_out) {
    if (!F || F->N < 0 || (F->N > 0 && (!F->keys_un || !F->desc))) return fail(ORBX_E_ARG, "bad frame view");
    if (F->N >= 65535) return fail(ORBX_E_ARG, "too many keypoints");
    if (!V || V->nlevels < 1 || V->nlevels > kMaxLevels || !V->scale_factors) return fail(ORBX_E_ARG, "bad scale levels");
    const int N = F->N;
    for (int i = 0; i < N; i++) assigned[i] = -1;
    if (nmatches_out) *nmatches_out = 0;
    if (M <= 0) return ORBX_OK;
    const size_t N1 = N > 0 ? N : 1, M1 = M;
    // input block: frame [keys | descriptors | uRight] | bad flags | (host points: descriptors | pos | normal | min | max)
    const size_t okp = 0, odesc = okp + al16(sizeof(KeyPointRec) * N1), our = odesc + al16(32 * N1), ob = our + al16(4 * N1), opd = ob + al16(M1),
                 opp = opd + (R ? 0 : al16(32 * M1)), opn = opp + (R ? 0 : al16(12 * M1)), opmn = opn + (R ? 0 : al16(12 * M1)), opmx = opmn + (R ? 0 : al16(4 * M1)),
                 in_total = opmx + (R ? 0 : al16(4 * M1));
    // output block: [counter 16 | start M | count M] | tracking fields | candidate entries
    const size_t hdr = al16(16 + 8 * M1), trk = track_block_bytes(M), oent = hdr + trk;
    const size_t pool = std::max<size_t>(h->area_pool, M1 * 48 + 1024);
    int e = h->h_packA.ensure(in_total + 16) | h->d_sr[SR_KPS].ensure(in_total + 16) | h->d_sr[SR_QUERY].ensure(sizeof(AreaQuery) * M1 + 16) |
            h->d_sr[SR_ENTRIES].ensure(oent + pool * 8 + 16) | h->h_out.ensure(oent + pool * 8 + 16);
    e |= h->d_si[SI_CELLOF].ensure(N + 1) | h->d_si[SI_CELLSTART].ensure(64 * 48 + 2) | h->d_si[SI_CELLITEMS].ensure(N + 1);
    if (e) return fail(ORBX_E_DEVICE, "upload/allocation failed");
    h->area_pool = pool;
    uint8_t* hp = h->h_packA.p;
    if (N > 0) { memcpy(hp + okp, F->keys_un, sizeof(KeyPointRec) * (size_t)N); memcpy(hp + odesc, F->desc, 32 * (size_t)N); }
    float* ur = (float*)(hp + our);
    if (F->u_right) memcpy(ur, F->u_right, sizeof(float) * (size_t)N); else for (size_t i = 0; i < N1; i++) ur[i] = -1.0f;
    if (is_bad) memcpy(hp + ob, is_bad, M1); else memset(hp + ob, 0, M1);
    if (!R) {
        memcpy(hp + opd, P->desc, 32 * M1); memcpy(hp + opp, P->pos, 12 * M1); memcpy(hp + opn, P->normal, 12 * M1);
        memcpy(hp + opmn, P->min_distance, 4 * M1); memcpy(hp + opmx, P->max_distance, 4 * M1);
    }
    if (rt::copy_h2d(h->d_sr[SR_KPS].p, hp, in_total, h->s0)) return fail(ORBX_E_DEVICE, "upload failed");
    const uint8_t* di = h->d_sr[SR_KPS].p;
    DeviceFrame D;
    D.kps = (const KeyPointRec*)(di + okp); D.desc = (const unsigned long long*)(di + odesc); D.ur = (const float*)(di + our);
    memset(&D.g, 0, sizeof D.g);
    D.g.min_x = F->min_x; D.g.min_y = F->min_y; D.g.gw_inv = F->grid_w_inv; D.g.gh_inv = F->grid_h_inv;
    D.cell_start = h->d_si[SI_CELLSTART].p; D.cell_items = h->d_si[SI_CELLITEMS].p;
    const dim3 one(1, 1, 1), blkg(kGridThreads, 1, 1), blk(256, 1, 1);
    ORBX_LAUNCH(k_grid_build, one, blkg, 0, h->s0, D.kps, N, D.g, h->d_si[SI_CELLOF].p, h->d_si[SI_CELLSTART].p, h->d_si[SI_CELLITEMS].p, (const int*)nullptr, 0);
    FrustumParams Fp; memset(&Fp, 0, sizeof Fp);
    memcpy(Fp.Rcw, V->Rcw, sizeof Fp.Rcw); memcpy(Fp.tcw, V->tcw, sizeof Fp.tcw); memcpy(Fp.Ow, V->Ow, sizeof Fp.Ow);
    memcpy(Fp.cam, V->cam, sizeof Fp.cam); Fp.kb8 = V->camera_type == 1;
    Fp.min_x = V->min_x; Fp.max_x = V->max_x; Fp.min_y = V->min_y; Fp.max_y = V->max_y; Fp.mbf = V->mbf; Fp.log_scale_factor = V->log_scale_factor; Fp.nlevels = V->nlevels;
    for (int l = 0; l < V->nlevels; l++) Fp.scale_factors[l] = V->scale_factors[l];
    Fp.cos_limit = cos_limit; Fp.th = th; Fp.th_far = th_far; Fp.far_points = far_points;
    uint8_t* dout = h->d_sr[SR_ENTRIES].p;
    int* d_counter = (int*)dout; int* d_start = (int*)(dout + 16); int* d_count = d_start + M; int2* d_ent = (int2*)(dout + oent);
    uint8_t* dtrk = dout + hdr;
    const size_t ol = al16(20 * M1), ov = ol + al16(4 * M1);                        // track block: [track 5 M][level M][in_view M] (scatter_track)
    AreaQuery* dq = (AreaQuery*)h->d_sr[SR_QUERY].p;
    const unsigned long long* dqd = R ? R->desc : (const unsigned long long*)(di + opd);
    {
        dim3 grid((M + 255) / 256, 1, 1);
        ORBX_LAUNCH(k_frustum, grid, blk, 0, h->s0, Fp, M, R ? R->pos : (const float*)(di + opp), R ? R->normal : (const float*)(di + opn),
                    R ? R->min_d : (const float*)(di + opmn), R ? R->max_d : (const float*)(di + opmx), (const uint8_t*)(di + ob),
                    dtrk + ov, (float*)dtrk, (int*)(dtrk + ol), dq, d_counter, (const FrustumParams*)nullptr);
    }
    {
        dim3 grid((M + kAreaWaves - 1) / kAreaWaves, 1, 1), blka(64 * kAreaWaves, 1, 1);
        ORBX_LAUNCH(k_area_search, grid, blka, 0, h->s0, (const AreaQuery*)dq, dqd, M, D.kps, D.ur, D.desc, D.g, D.cell_start, D.cell_items, 1, d_counter, (int)pool,
                    d_start, d_count, d_ent, 0);
    }
    const size_t guess = std::min(pool, std::max<size_t>(h->area_last_total + h->area_last_total / 4 + 256, 1024));
    rt::copy_d2h(h->h_out.p, dout, oent + guess * 8, h->s0);
    if (rt::stream_sync(h->s0) || rt::check_launch()) return fail(ORBX_E_DEVICE, "local point search failed: %s", rt::last_error());
    const int total = *(const int*)h->h_out.p;
    if ((size_t)total > pool) {                                     // pool overflow (the first call of a much denser scene): grow and run again
        h->area_pool = (size_t)total + 1024;
        return search_local_points_core(h, F, V, M, P, R, is_bad, has_obs, cos_limit, th, far_points, th_far, nnratio, out, assigned, nmatches_out);
    }
    if ((size_t)total > guess) {
        rt::copy_d2h(h->h_out.p + oent + guess * 8, dout + oent + guess * 8, ((size_t)total - guess) * 8, h->s0);
        if (rt::stream_sync(h->s0)) return fail(ORBX_E_DEVICE, "local point search download failed: %s", rt::last_error());
    }
    h->area_last_total = (size_t)total;
    scatter_track(h->h_out.p + hdr, M, out);
    const int* q_start = (const int*)(h->h_out.p + 16); const int* q_count = q_start + M; const int* ent = (const int*)(h->h_out.p + oent);
    // ---- sequential replay of src/ORBmatcher.cc:62-166 (a query without candidates - not in view, far, bad - has count 0) ----
    std::vector<uint8_t> occ(N1, 0);
    if (F->occupied) memcpy(occ.data(), F->occupied, N);
    int nmatches = 0;
    for (int i = 0; i < M; i++) {
        if (q_count[i] == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < q_count[i]; k++) {
            const int idx = ent[2 * (size_t)(q_start[i] + k)], dl = ent[2 * (size_t)(q_start[i] + k) + 1];
            if (occ[idx]) continue;
            const int dist = dl & 0xFFFF, level = dl >> 16;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = level; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = level; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                assigned[bestIdx] = i;
                occ[bestIdx] = has_obs ? has_obs[i] : 1;
                nmatches++;
            }
        }
    }
    if (nmatches_out) *nmatches_out = nmatches;
    return ORBX_OK;
}
}  // namespace

int orbm_search_local_points(orbx_extractor* h, const OrbmFrameView* F, const OrbmFrustumView* V, const OrbmWorldPointView* P, float cos_limit, float th,
                             int far_points, float th_far, float nnratio, const OrbmTrackOut* out, int* assigned, int* nmatches_out) {
    if (!h || !F || !V || !P || !assigned) return fail(ORBX_E_ARG, "null");
    if (P->M < 0 || (P->M > 0 && (!P->pos || !P->normal || !P->min_distance || !P->max_distance || !P->desc))) return fail(ORBX_E_ARG, "bad map point view");
    rt::set_device(h->device);
    return search_local_points_core(h, F, V, P->M, P, nullptr, P->is_bad, P->has_obs, cos_limit, th, far_points, th_far, nnratio, out, assigned, nmatches_out);
}

int orbm_search_local_points_resident(orbx_extractor* h, const OrbmFrameView* F, const OrbmFrustumView* V, const orbm_points* points, const uint8_t* is_bad,
                                      const uint8_t* has_obs, float cos_limit, float th, int far_points, float th_far, float nnratio, const OrbmTrackOut* out,
                                      int* assigned, int* nmatches_out) {
    if (!h || !F || !V || !points || !assigned) return fail(ORBX_E_ARG, "null");
    if (points->device != h->device) return fail(ORBX_E_ARG, "map points live on another device");
    rt::set_device(h->device);
    return search_local_points_core(h, F, V, points->M, nullptr, points, is_bad, has_obs, cos_limit, th, far_points, th_far, nnratio, out, assigned, nmatches_out);
}

// ---- the batched, device-resident form of Tracking::SearchLocalPoints ----------------------------------------------
// Frames = images [first, first + B) of the handle's last extraction, read where the extractor left them (keypoints, descriptors, counts; uRight
// from orbm_stereo_match / orbm_stereo_from_depth); map points resident (orbm_points); one pose per frame.  Five launches for the whole batch:
// grid build (B workgroups), frustum + window queries (B x M threads), window search (B x M waves), accept loop (one wave per frame) - nothing
// visits the host between them; one upload (poses + call-time flags), one download (assignments, match counts, optionally mbTrackInView).

int orbm_stereo_from_depth(orbx_extractor* h, int first, int B, const float* depth, int stride, size_t image_stride, int on_device, float mbf) {
    if (!h || !depth || B <= 0 || first < 0 || first + B > h->lastB) return fail(ORBX_E_ARG, "bad frame range");
    if (stride < h->W || (B > 1 && image_stride < (size_t)stride * (h->H - 1) + h->W)) return fail(ORBX_E_ARG, "depth stride too small");
    if (undistort_stale(h)) return fail(ORBX_E_ARG, "orbx_set_undistort was called after the last extraction: extract again before orbm_stereo_from_depth");
    rt::set_device(h->device);
    const float* d_depth = depth;
    if (!on_device) {
        const size_t n = (size_t)(B - 1) * image_stride + (size_t)stride * (h->H - 1) + h->W;
        if (h->d_depth_in.ensure(n * sizeof(float) + 16) || rt::copy_h2d(h->d_depth_in.p, depth, n * sizeof(float), h->s0)) return fail(ORBX_E_DEVICE, "depth upload failed");
        d_depth = (const float*)h->d_depth_in.p;
    }
    const int cap = h->kp_total_cap;
    rt::memset_async(h->d_nmatch.p, 0, sizeof(int) * (size_t)B, h->s0);
    dim3 grid((cap + 255) / 256, B, 1), blk(256, 1, 1);
    ORBX_LAUNCH(k_stereo_from_depth, grid, blk, 0, h->s0, (const KeyPointRec*)(h->d_kps.p + (size_t)first * cap),
                h->ex_undist_active ? (const KeyPointRec*)(h->d_kps_un.p + (size_t)first * cap) : (const KeyPointRec*)nullptr, (const int*)(h->d_nm.p + first), cap,
                d_depth, stride, image_stride, h->W, h->H, mbf, h->d_uRight.p, h->d_depth.p, h->d_nmatch.p);
    if (rt::check_launch()) return fail(ORBX_E_DEVICE, "kernel launch failed: %s", rt::last_error());
    return ORBX_OK;
}

int orbm_search_local_points_batch(orbx_extractor* h, int first, int B, const OrbmFrustumView* frames, const orbm_points* points, const uint8_t* is_bad,
                                   const uint8_t* has_obs, const uint8_t* occupied, int use_u_right, float cos_limit, float th, int far_points, float th_far,
                                   float nnratio, int want_in_view) {
    if (h) h->lp_B = 0;            // a new enqueue - accepted or refused - ends the previous batch: a refused call leaves nothing to fetch
    if (!h || !frames || !points || B <= 0 || first < 0 || first + B > h->lastB) return fail(ORBX_E_ARG, "bad frame range / null");
    if (points->device != h->device) return fail(ORBX_E_ARG, "map points live on another device");
    for (int b = 0; b < B; b++) {
        if (frames[b].nlevels < 1 || frames[b].nlevels > kMaxLevels || !frames[b].scale_factors) return fail(ORBX_E_ARG, "bad scale levels (frame %d)", b);
        // one grid geometry per batch (the frames of a batch come from one extraction: one image size)
        if (frames[b].min_x != frames[0].min_x || frames[b].max_x != frames[0].max_x || frames[b].min_y != frames[0].min_y || frames[b].max_y != frames[0].max_y)
            return fail(ORBX_E_ARG, "frame %d has other image bounds than frame 0", b);
    }
    if (!(frames[0].max_x > frames[0].min_x) || !(frames[0].max_y > frames[0].min_y)) return fail(ORBX_E_ARG, "empty image bounds");
    if (undistort_stale(h)) return fail(ORBX_E_ARG, "orbx_set_undistort was called after the last extraction: extract again before searching its frames");
    rt::set_device(h->device);
    const int M = points->M, cap = h->kp_total_cap;
    const size_t B1 = B, M1 = M > 0 ? M : 1, C1 = cap;
    // every refusal comes BEFORE anything is enqueued into the shared block, and leaves no batch to fetch
    const size_t smem_accept = 4 * (size_t)((cap + 31) / 32) + 4 * (size_t)cap + 64;
    if (smem_accept + 1024 > rt::lds_limit(h->device)) { h->lp_B = 0; return fail(ORBX_E_CAPACITY, "%d keypoints per frame need %zu bytes of LDS in the accept kernel", cap, smem_accept); }
    h->lp_B = 0;                                                // the block is about to be overwritten: a previous, unfetched batch is gone
    if (h->lp_pending) { rt::event_sync(h->ev_lp); }            // the staging block of the previous enqueue has been consumed
    // upload block: poses | bad flags | has-observation flags | occupancy
    const size_t u_f = 0, u_bad = u_f + al16(sizeof(FrustumParams) * B1), u_obs = u_bad + al16(M1), u_occ = u_obs + al16(M1), u_total = u_occ + (occupied ? al16(B1 * C1) : 0);
    // device block: upload | uRight of "no stereo" | grid (cell_of, cell_start, cell_items) | queries | track | level | in_view | q_start | q_count |
    // result [counter 16 | nmatches B | assigned B * cap] | entry pool
    size_t o = al16(u_total);
    const size_t o_ur = o; o += use_u_right ? 0 : al16(4 * B1 * C1);
    const size_t o_cof = o; o += al16(4 * B1 * C1);
    const size_t o_cst = o; o += al16(4 * B1 * kGridCellStride);
    const size_t o_cit = o; o += al16(4 * B1 * C1);
    const size_t o_q = o; o += al16(sizeof(AreaQuery) * B1 * M1);
    const size_t o_trk = o; o += al16(20 * B1 * M1);
    const size_t o_lvl = o; o += al16(4 * B1 * M1);
    const size_t o_view = o; o += al16(B1 * M1);
    const size_t o_qs = o; o += al16(4 * B1 * M1);
    const size_t o_qc = o; o += al16(4 * B1 * M1);
    const size_t o_res = o; const size_t res_bytes = 16 + al16(4 * B1) + 4 * B1 * C1; o += al16(res_bytes);
    const size_t o_pool = o;
    size_t pool = std::max<size_t>(h->lp_pool, B1 * M1 * 6 + 4096);
    if (pool > 0x7fffffff / 2) pool = 0x7fffffff / 2;
    if (h->d_lp.ensure(o_pool + pool * 8 + 64) || h->h_lp_in.ensure(u_total + 16) || h->h_lp_out.ensure(al16(res_bytes) + (want_in_view ? B1 * M1 : 0) + 64))
        return fail(ORBX_E_DEVICE, "allocation failed (batched local point search, %d frames x %d points)", B, M);
    h->lp_pool = pool;
    uint8_t* hp = h->h_lp_in.p; uint8_t* dp = h->d_lp.p;
    FrustumParams* Fp = (FrustumParams*)(hp + u_f);
    for (int b = 0; b < B; b++) fill_frustum_params(&frames[b], cos_limit, th, far_points, th_far, &Fp[b]);
    if (is_bad) memcpy(hp + u_bad, is_bad, M1); else memset(hp + u_bad, 0, M1);
    if (has_obs) memcpy(hp + u_obs, has_obs, M1); else memset(hp + u_obs, 1, M1);
    if (occupied) memcpy(hp + u_occ, occupied, B1 * C1);
    if (rt::copy_h2d(dp, hp, u_total, h->s0) || rt::event_record(h->ev_lp, h->s0)) return fail(ORBX_E_DEVICE, "upload failed: %s", rt::last_error());
    h->lp_pending = true;
    const KeyPointRec* kps = (h->ex_undist_active ? h->d_kps_un.p : h->d_kps.p) + (size_t)first * cap;     // mvKeysUn
    const unsigned long long* fdesc = h->d_desc.p + (size_t)first * cap * 4;
    const int* nper = h->d_nm.p + first;
    const float* ur = h->d_uRight.p;
    if (!use_u_right) { rt::memset_async(dp + o_ur, 0xBF, 4 * B1 * C1, h->s0); ur = (const float*)(dp + o_ur); }   // 0xBFBFBFBF = -1.498...: a negative uRight = monocular keypoint
    GridParams g; memset(&g, 0, sizeof g);
    g.min_x = frames[0].min_x; g.min_y = frames[0].min_y;
    g.gw_inv = (float)kGridColsHost / (frames[0].max_x - frames[0].min_x); g.gh_inv = (float)kGridRowsHost / (frames[0].max_y - frames[0].min_y);   // src/Frame.cc:190-191
    int* d_counter = (int*)(dp + o_res); int* d_nmatch = (int*)(dp + o_res + 16); int* d_assigned = (int*)(dp + o_res + 16 + al16(4 * B1));
    if (h->profile) rt::event_record(h->ev_stage[ST_MATCH][0], h->s0);
    {
        dim3 grid(B, 1, 1), blkg(kGridThreads, 1, 1);
        ORBX_LAUNCH(k_grid_build, grid, blkg, 0, h->s0, kps, 0, g, (int*)(dp + o_cof), (int*)(dp + o_cst), (int*)(dp + o_cit), nper, cap);
    }
    if (M > 0) {
        FrustumParams dummy; memset(&dummy, 0, sizeof dummy);
        dim3 grid((M + 255) / 256, B, 1), blk(256, 1, 1);
        ORBX_LAUNCH(k_frustum, grid, blk, 0, h->s0, dummy, M, points->pos, points->normal, points->min_d, points->max_d, (const uint8_t*)(dp + u_bad),
                    dp + o_view, (float*)(dp + o_trk), (int*)(dp + o_lvl), (AreaQuery*)(dp + o_q), d_counter, (const FrustumParams*)(dp + u_f));
        dim3 grida((M + 255) / 256, B, 1);
        ORBX_LAUNCH(k_area_search_threads, grida, blk, 0, h->s0, (const AreaQuery*)(dp + o_q), points->desc, M, kps, ur, fdesc, g, (const int*)(dp + o_cst), (const int*)(dp + o_cit), 1,
                    d_counter, (int)pool, (int*)(dp + o_qs), (int*)(dp + o_qc), (int2*)(dp + o_pool), cap, 0);
    } else rt::memset_async(d_counter, 0, 16, h->s0);
    {
        dim3 grid(B, 1, 1), blk(64, 1, 1);
        const size_t smem = smem_accept;
        ORBX_LAUNCH(k_local_accept, grid, blk, smem, h->s0, M, cap, nper, (const int*)(dp + o_qs), (const int*)(dp + o_qc), (const int2*)(dp + o_pool),
                    occupied ? (const uint8_t*)(dp + u_occ) : (const uint8_t*)nullptr, (const uint8_t*)(dp + u_obs), nnratio, TH_HIGH, d_assigned, d_nmatch);
    }
    if (h->profile) rt::event_record(h->ev_stage[ST_MATCH][1], h->s0);
    if (rt::check_launch()) return fail(ORBX_E_DEVICE, "kernel launch failed: %s", rt::last_error());
    h->lp_B = B; h->lp_M = M; h->lp_first = first; h->lp_o_counter = o_res; h->lp_o_view = o_view; h->lp_want_view = want_in_view != 0;
    return ORBX_OK;
}

namespace {
// one record per frame of map points seen from somewhere else: the LastFrame's (octave of its keypoint, has_obs) or a key frame's (distance limits)
struct PointRows {
    int cap; const int* n; const float* pos; const uint8_t* valid; const int* octave; const float* angle; const uint8_t* has_obs; const uint8_t* desc;
    const float* min_distance; const float* max_distance;          // key-frame form only
};
// SearchByProjection(CurrentFrame, LastFrame) (keyframe = false) or (CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (keyframe = true) for B frames
int projection_batch(orbx_extractor* h, int first, int B, const OrbmFrustumView* cur, const PointRows* last, float th, const uint8_t* forward, const uint8_t* backward,
                     int check_ori, const uint8_t* occupied, int use_u_right, bool keyframe, int th_accept) {
    const int M = last->cap;
    for (int b = 0; b < B; b++) {
        if (cur[b].nlevels < 1 || cur[b].nlevels > kMaxLevels || !cur[b].scale_factors) return fail(ORBX_E_ARG, "bad scale levels (frame %d)", b);
        if (cur[b].min_x != cur[0].min_x || cur[b].max_x != cur[0].max_x || cur[b].min_y != cur[0].min_y || cur[b].max_y != cur[0].max_y)
            return fail(ORBX_E_ARG, "frame %d has other image bounds than frame 0", b);
        if (last->n[b] < 0 || last->n[b] > M) return fail(ORBX_E_ARG, "frame %d: %d points in %d rows", b, last->n[b], M);
    }
    if (!(cur[0].max_x > cur[0].min_x) || !(cur[0].max_y > cur[0].min_y)) return fail(ORBX_E_ARG, "empty image bounds");
    if (undistort_stale(h)) return fail(ORBX_E_ARG, "orbx_set_undistort was called after the last extraction: extract again before searching its frames");
    rt::set_device(h->device);
    const int cap = h->kp_total_cap;
    const size_t B1 = B, M1 = M, C1 = cap;
    const size_t smem_accept = 4 * (size_t)((cap + 31) / 32) + 4 * (size_t)cap + 4 * (size_t)M + 64;          // grows with cap_last: checked before anything is enqueued
    if (smem_accept + 1024 > rt::lds_limit(h->device)) { h->lp_B = 0; return fail(ORBX_E_CAPACITY, "%d keypoints and %d last-frame points per frame need %zu bytes of LDS in the accept kernel", cap, M, smem_accept); }
    h->lp_B = 0;
    if (h->lp_pending) rt::event_sync(h->ev_lp);
    // upload block: poses | n_last | pos | valid | octave | angle | has_obs | descriptors | occupancy
    const size_t u_f = 0, u_n = u_f + al16(sizeof(FrustumParams) * B1), u_pos = u_n + al16(4 * B1), u_val = u_pos + al16(12 * B1 * M1), u_oct = u_val + al16(B1 * M1),
                 u_ang = u_oct + al16(4 * B1 * M1), u_obs = u_ang + al16(4 * B1 * M1), u_desc = u_obs + al16(B1 * M1), u_occ = u_desc + al16(32 * B1 * M1),
                 u_mxd = u_occ + (occupied ? al16(B1 * C1) : 0), u_total = u_mxd + (keyframe ? al16(4 * B1 * M1) : 0);
    size_t o = al16(u_total);
    const size_t o_ur = o; o += use_u_right ? 0 : al16(4 * B1 * C1);
    const size_t o_cof = o; o += al16(4 * B1 * C1);
    const size_t o_cst = o; o += al16(4 * B1 * kGridCellStride);
    const size_t o_cit = o; o += al16(4 * B1 * C1);
    const size_t o_q = o; o += al16(sizeof(AreaQuery) * B1 * M1);
    const size_t o_qs = o; o += al16(4 * B1 * M1);
    const size_t o_qc = o; o += al16(4 * B1 * M1);
    const size_t o_res = o; const size_t res_bytes = 16 + al16(4 * B1) + 4 * B1 * C1; o += al16(res_bytes);
    const size_t o_pool = o;
    size_t pool = std::max<size_t>(h->lp_pool, B1 * M1 * 16 + 4096);           // th = 7 .. 15 px windows: a dozen candidates per point
    if (pool > 0x7fffffff / 2) pool = 0x7fffffff / 2;
    if (h->d_lp.ensure(o_pool + pool * 8 + 64) || h->h_lp_in.ensure(u_total + 16) || h->h_lp_out.ensure(al16(res_bytes) + 64))
        return fail(ORBX_E_DEVICE, "allocation failed (batched last-frame search, %d frames x %d points)", B, M);
    h->lp_pool = pool;
    uint8_t* hp = h->h_lp_in.p; uint8_t* dp = h->d_lp.p;
    FrustumParams* Fp = (FrustumParams*)(hp + u_f);
    for (int b = 0; b < B; b++) {
        fill_frustum_params(&cur[b], 0.0f, th, 0, 0.0f, &Fp[b]);
        Fp[b].forward = forward ? forward[b] != 0 : 0; Fp[b].backward = backward ? backward[b] != 0 : 0; Fp[b].debug_flags = h->debug_stereo_flags;
    }
    memcpy(hp + u_n, last->n, 4 * B1); memcpy(hp + u_pos, last->pos, 12 * B1 * M1); memcpy(hp + u_val, last->valid, B1 * M1);
    if (keyframe) { memcpy(hp + u_oct, last->min_distance, 4 * B1 * M1); memcpy(hp + u_mxd, last->max_distance, 4 * B1 * M1); }     // (the octave rows carry mfMinDistance)
    else memcpy(hp + u_oct, last->octave, 4 * B1 * M1);
    memcpy(hp + u_ang, last->angle, 4 * B1 * M1);
    if (last->has_obs) memcpy(hp + u_obs, last->has_obs, B1 * M1); else memset(hp + u_obs, 1, B1 * M1);
    memcpy(hp + u_desc, last->desc, 32 * B1 * M1);
    if (occupied) memcpy(hp + u_occ, occupied, B1 * C1);
    if (rt::copy_h2d(dp, hp, u_total, h->s0) || rt::event_record(h->ev_lp, h->s0)) return fail(ORBX_E_DEVICE, "upload failed: %s", rt::last_error());
    h->lp_pending = true;
    const KeyPointRec* kps = (h->ex_undist_active ? h->d_kps_un.p : h->d_kps.p) + (size_t)first * cap;
    const unsigned long long* fdesc = h->d_desc.p + (size_t)first * cap * 4;
    const int* nper = h->d_nm.p + first;
    const float* ur = h->d_uRight.p;
    if (!use_u_right) { rt::memset_async(dp + o_ur, 0xBF, 4 * B1 * C1, h->s0); ur = (const float*)(dp + o_ur); }
    GridParams g; memset(&g, 0, sizeof g);
    g.min_x = cur[0].min_x; g.min_y = cur[0].min_y;
    g.gw_inv = (float)kGridColsHost / (cur[0].max_x - cur[0].min_x); g.gh_inv = (float)kGridRowsHost / (cur[0].max_y - cur[0].min_y);
    int* d_counter = (int*)(dp + o_res); int* d_nmatch = (int*)(dp + o_res + 16); int* d_assigned = (int*)(dp + o_res + 16 + al16(4 * B1));
    if (h->profile) rt::event_record(h->ev_stage[ST_MATCH][0], h->s0);
    const dim3 blk(256, 1, 1);
    {
        dim3 grid(B, 1, 1), blkg(kGridThreads, 1, 1);
        ORBX_LAUNCH(k_grid_build, grid, blkg, 0, h->s0, kps, 0, g, (int*)(dp + o_cof), (int*)(dp + o_cst), (int*)(dp + o_cit), nper, cap);
    }
    {
        dim3 grid((M + 255) / 256, B, 1);
        if (keyframe) ORBX_LAUNCH(k_keyframe_queries, grid, blk, 0, h->s0, (const FrustumParams*)(dp + u_f), M, (const int*)(dp + u_n), (const float*)(dp + u_pos),
                                  (const uint8_t*)(dp + u_val), (const float*)(dp + u_oct), (const float*)(dp + u_mxd), (AreaQuery*)(dp + o_q), d_counter);
        else ORBX_LAUNCH(k_lastframe_queries, grid, blk, 0, h->s0, (const FrustumParams*)(dp + u_f), M, (const int*)(dp + u_n), (const float*)(dp + u_pos), (const uint8_t*)(dp + u_val),
                    (const int*)(dp + u_oct), (AreaQuery*)(dp + o_q), d_counter);
        ORBX_LAUNCH(k_area_search_threads, grid, blk, 0, h->s0, (const AreaQuery*)(dp + o_q), (const unsigned long long*)(dp + u_desc), M, kps, ur, fdesc, g, (const int*)(dp + o_cst),
                    (const int*)(dp + o_cit), keyframe ? 0 : 1, d_counter, (int)pool, (int*)(dp + o_qs), (int*)(dp + o_qc), (int2*)(dp + o_pool), cap, 1);
    }
    {
        dim3 grid(B, 1, 1), blka(64, 1, 1);
        const size_t smem = smem_accept;
        ORBX_LAUNCH(k_lastframe_accept, grid, blka, smem, h->s0, M, cap, nper, (const int*)(dp + o_qs), (const int*)(dp + o_qc), (const int2*)(dp + o_pool),
                    occupied ? (const uint8_t*)(dp + u_occ) : (const uint8_t*)nullptr, (const uint8_t*)(dp + u_obs), th_accept, d_assigned, d_nmatch,
                    (const float*)(dp + u_ang), kps, check_ori);
    }
    if (h->profile) rt::event_record(h->ev_stage[ST_MATCH][1], h->s0);
    if (rt::check_launch()) return fail(ORBX_E_DEVICE, "kernel launch failed: %s", rt::last_error());
    h->lp_B = B; h->lp_M = M; h->lp_first = first; h->lp_o_counter = o_res; h->lp_o_view = 0; h->lp_want_view = false;
    return ORBX_OK;
}
}  // namespace

int orbm_search_by_projection_lastframe_batch(orbx_extractor* h, int first, int B, const OrbmFrustumView* cur, const OrbmLastFrameBatch* last, float th,
                                              const uint8_t* forward, const uint8_t* backward, int check_ori, const uint8_t* occupied, int use_u_right) {
    if (h) h->lp_B = 0;
    if (!h || !cur || !last || B <= 0 || first < 0 || first + B > h->lastB) return fail(ORBX_E_ARG, "bad frame range / null");
    if (last->cap_last <= 0 || !last->n || !last->pos || !last->valid || !last->octave || !last->angle || !last->desc) return fail(ORBX_E_ARG, "bad last-frame batch");
    const PointRows R = {last->cap_last, last->n, last->pos, last->valid, last->octave, last->angle, last->has_obs, last->desc, nullptr, nullptr};
    return projection_batch(h, first, B, cur, &R, th, forward, backward, check_ori, occupied, use_u_right, false, TH_HIGH);
}

int orbm_search_by_projection_keyframe_batch(orbx_extractor* h, int first, int B, const OrbmFrustumView* cur, const OrbmKeyFramePointBatch* kf, float th, int orb_dist,
                                             int check_ori, const uint8_t* occupied) {
    if (h) h->lp_B = 0;
    if (!h || !cur || !kf || B <= 0 || first < 0 || first + B > h->lastB) return fail(ORBX_E_ARG, "bad frame range / null");
    if (kf->cap_kf <= 0 || !kf->n || !kf->pos || !kf->valid || !kf->min_distance || !kf->max_distance || !kf->angle || !kf->desc) return fail(ORBX_E_ARG, "bad key-frame batch");
    if (orb_dist < 0 || orb_dist > 256) return fail(ORBX_E_ARG, "ORBdist out of range");
    // every accepted point occupies its keypoint (`if(CurrentFrame.mvpMapPoints[i2]) continue;`, :2259-2260): has_obs = all, and no right-coordinate gate
    const PointRows R = {kf->cap_kf, kf->n, kf->pos, kf->valid, nullptr, kf->angle, nullptr, kf->desc, kf->min_distance, kf->max_distance};
    return projection_batch(h, first, B, cur, &R, th, nullptr, nullptr, check_ori, occupied, /*use_u_right*/ 0, true, orb_dist);
}

int orbm_search_local_points_fetch(orbx_extractor* h, int* assigned, int cap, int* nmatches, uint8_t* in_view) {
    if (!h || h->lp_B <= 0) return fail(ORBX_E_ARG, "no batched local point search is pending");
    if (assigned && cap < h->kp_total_cap) return fail(ORBX_E_CAPACITY, "assigned rows need %d entries", h->kp_total_cap);
    rt::set_device(h->device);
    const size_t B1 = h->lp_B, M1 = h->lp_M > 0 ? h->lp_M : 1, C1 = h->kp_total_cap;
    const size_t res_bytes = 16 + al16(4 * B1) + 4 * B1 * C1;
    uint8_t* hp = h->h_lp_out.p;
    int e = rt::copy_d2h(hp, h->d_lp.p + h->lp_o_counter, res_bytes, h->s0);
    if (in_view && h->lp_want_view && h->lp_M > 0) e |= rt::copy_d2h(hp + al16(res_bytes), h->d_lp.p + h->lp_o_view, B1 * M1, h->s0);
    if (e || rt::stream_sync(h->s0) || rt::check_launch()) return fail(ORBX_E_DEVICE, "batched local point search failed: %s", rt::last_error());
    h->lp_pending = false;
    if (h->profile) h->stage_ms[ST_MATCH] = rt::event_elapsed_ms(h->ev_stage[ST_MATCH][0], h->ev_stage[ST_MATCH][1]);
    const int total = *(const int*)hp;
    if (h->lp_M > 0 && (size_t)total > h->lp_pool) {                // the candidate pool was too small: the caller enqueues again (the pool has grown)
        h->lp_pool = (size_t)total + (size_t)total / 8 + 4096;
        return fail(ORBX_E_CAPACITY, "candidate pool overflow (%d entries): enqueue the search again, the pool has been enlarged", total);
    }
    if (nmatches) memcpy(nmatches, hp + 16, 4 * B1);
    if (assigned) {
        const int* src = (const int*)(hp + 16 + al16(4 * B1));
        if ((size_t)cap == C1) memcpy(assigned, src, 4 * B1 * C1);
        else for (size_t b = 0; b < B1; b++) memcpy(assigned + b * (size_t)cap, src + b * C1, 4 * C1);
    }
    if (in_view) { if (h->lp_want_view && h->lp_M > 0) memcpy(in_view, hp + al16(res_bytes), B1 * M1); else return fail(ORBX_E_ARG, "in_view was not requested at enqueue time"); }
    return ORBX_OK;
}

int orbm_search_by_projection_frame(orbx_extractor* h, const OrbmFrameView* Cur, const OrbmLastFrameView* Last, float th, int forward, int backward,
                                    int check_ori, int* assigned, int* nmatches_out) {
    if (!h || !Cur || !Last || !assigned) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    DeviceFrame D;
    int rc = upload_frame(h, Cur, &D); if (rc) return rc;
    const int NL = Last->N, N = Cur->N;
    std::vector<AreaQuery> qs(NL);
    for (int i = 0; i < NL; i++) {
        AreaQuery& q = qs[i]; memset(&q, 0, sizeof q);
        if (!Last->valid[i]) continue;
        const float u = Last->proj_u[i], v = Last->proj_v[i];
        if (u < Cur->min_x || u > Cur->max_x) continue;                  // :2003-2006
        if (v < Cur->min_y || v > Cur->max_y) continue;
        const int oct = Last->octave[i];
        if (oct < 0 || oct >= Cur->nlevels) continue;
        q.x = u; q.y = v; q.r = th * Cur->scale_factors[oct];            // :2012
        q.ur = u - Cur->mbf * Last->inv_z[i];                            // :2045
        if (forward) { q.min_level = oct; q.max_level = -1; }            // :2018-2023
        else if (backward) { q.min_level = 0; q.max_level = oct; }
        else { q.min_level = oct - 1; q.max_level = oct + 1; }
        q.active = 1; q.gate = 1;
    }
    Csr c;
    rc = run_area_search(h, D, qs, Last->desc, &c); if (rc) return rc;
    std::vector<uint8_t> occ(N > 0 ? N : 1, 0);
    if (Cur->occupied) memcpy(occ.data(), Cur->occupied, N);
    for (int i = 0; i < N; i++) assigned[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    for (int i = 0; i < NL; i++) {
        if (!qs[i].active || c.count[i] == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < c.count[i]; k++) {
            const int i2 = c.ent[2 * (size_t)(c.start[i] + k)], dist = c.ent[2 * (size_t)(c.start[i] + k) + 1] & 0xFFFF;
            if (occ[i2]) continue;
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            assigned[bestIdx2] = i;
            occ[bestIdx2] = Last->has_obs ? Last->has_obs[i] : 1;
            nmatches++;
            if (check_ori) rotHist[rot_bin(Last->angle[i], Cur->keys_un[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { assigned[idx] = -2; nmatches--; }
    }
    if (nmatches_out) *nmatches_out = nmatches;
    return ORBX_OK;
}

// ORBmatcher::SearchForTriangulation for one key frame against n2 neighbours in a single launch (LocalMapping::CreateNewMapPoints calls
// it for 10-30 neighbours in a row, src/LocalMapping.cc:510-540).  matches12: n2 rows of K1->N entries.
static int search_for_triangulation_impl(orbx_extractor* h, const OrbmKeyFrameView* K1, int n2, const OrbmKeyFrameView* const* K2s, const float* F12s,
                                         const float* eps, const OrbmKB8Pair* kb8, int only_stereo, int coarse, int check_ori, int* matches12, int* nmatches_out) {
    if (!h || !K1 || n2 < 0 || (n2 > 0 && (!K2s || (!F12s && !kb8) || !eps || !matches12))) return fail(ORBX_E_ARG, "null");
    if (K1->N >= 65535) return fail(ORBX_E_ARG, "keyframe too large");
    rt::set_device(h->device);
    const int N1 = K1->N;
    for (size_t i = 0; i < (size_t)n2 * N1; i++) matches12[i] = -1;
    for (int j = 0; j < n2; j++) if (nmatches_out) nmatches_out[j] = 0;
    // merge-join of the feature vectors by node id (:1105-1286); one work item per unmatched feature of KF1 and neighbour.
    // The neighbours' arrays are concatenated; feat2 entries and the returned indices are global (base2[j] + index in KF j).
    std::vector<BowItem> items;
    std::vector<int> base2(n2 + 1, 0), fbase(n2 + 1, 0), first_item(n2 + 1, 0);
    for (int j = 0; j < n2; j++) {
        const OrbmKeyFrameView* K2 = K2s[j];
        if (!K2 || K2->N >= 65535 || K2->nlevels > kMaxLevels) return fail(ORBX_E_ARG, "bad neighbour key frame %d", j);
        base2[j + 1] = base2[j] + K2->N; fbase[j + 1] = fbase[j] + K2->fv_start[K2->fv_nodes];
        first_item[j] = (int)items.size();
        int a = 0, b = 0;
        while (a < K1->fv_nodes && b < K2->fv_nodes) {
            const uint32_t na = K1->fv_node_id[a], nb = K2->fv_node_id[b];
            if (na == nb) {
                for (int k = K1->fv_start[a]; k < K1->fv_start[a + 1]; k++) {
                    const int idx1 = (int)K1->fv_feat[k];
                    if (K1->has_map_point && K1->has_map_point[idx1]) continue;
                    const bool stereo1 = K1->u_right && K1->u_right[idx1] >= 0;
                    if (only_stereo && !stereo1) continue;
                    BowItem it; it.idx1 = idx1; it.start2 = fbase[j] + K2->fv_start[b]; it.cnt2 = K2->fv_start[b + 1] - K2->fv_start[b]; it.out_off = j;
                    if (it.cnt2 > 0xFFFF) return fail(ORBX_E_ARG, "vocabulary node with more than 65535 features");
                    items.push_back(it);
                }
                a++; b++;
            } else if (na < nb) { while (a < K1->fv_nodes && K1->fv_node_id[a] < nb) a++; }
            else { while (b < K2->fv_nodes && K2->fv_node_id[b] < na) b++; }
        }
    }
    first_item[n2] = (int)items.size();
    if (items.empty()) return ORBX_OK;
    const int T2 = std::max(base2[n2], 1), TF = std::max(fbase[n2], 1);
    std::vector<float> ur1(N1 > 0 ? N1 : 1, -1.0f), ur2(T2, -1.0f);
    if (K1->u_right) memcpy(ur1.data(), K1->u_right, sizeof(float) * N1);
    std::vector<uint8_t> mp2(T2, 0), desc2((size_t)T2 * 32);
    std::vector<KeyPointRec> kps2(T2);
    std::vector<int> feat2(TF);
    std::vector<BowParams> Ps(n2);
    for (int j = 0; j < n2; j++) {
        const OrbmKeyFrameView* K2 = K2s[j];
        const int N2 = K2->N, o = base2[j];
        if (N2 > 0) { memcpy(&kps2[o], K2->keys_un, sizeof(KeyPointRec) * (size_t)N2); memcpy(&desc2[(size_t)o * 32], K2->desc, 32 * (size_t)N2); }
        if (K2->u_right) memcpy(&ur2[o], K2->u_right, sizeof(float) * (size_t)N2);
        if (K2->has_map_point) memcpy(&mp2[o], K2->has_map_point, N2);
        const int nf = K2->fv_start[K2->fv_nodes];
        for (int k = 0; k < nf; k++) feat2[fbase[j] + k] = (int)K2->fv_feat[k] + o;
        BowParams& P = Ps[j]; memset(&P, 0, sizeof P);
        if (F12s) for (int i = 0; i < 9; i++) P.F12[i] = F12s[9 * (size_t)j + i];
        if (kb8) {                               // Kannala-Brandt cameras: one OrbmKB8Pair per neighbour
            const OrbmKB8Pair& C = kb8[j];
            P.kb8 = 1; P.nleft1 = C.nleft1; P.nleft2 = C.nleft2;
            memcpy(P.cam1, C.cam1, sizeof P.cam1); memcpy(P.cam2, C.cam2, sizeof P.cam2); memcpy(P.R, C.R, sizeof P.R); memcpy(P.t, C.t, sizeof P.t);
            for (int l = 0; l < K1->nlevels && l < kMaxLevels; l++) P.sigma2_1[l] = K1->level_sigma2[l];
        }
        P.ep[0] = eps[2 * (size_t)j]; P.ep[1] = eps[2 * (size_t)j + 1];
        for (int l = 0; l < K2->nlevels; l++) { P.scale2[l] = K2->scale_factors[l]; P.sigma2_2[l] = K2->level_sigma2[l]; }
        P.only_stereo = only_stereo; P.coarse = coarse; P.th_low = TH_LOW;
    }
    Packer pk(h);
    const size_t pk1 = pk.add(K1->keys_un, sizeof(KeyPointRec) * (size_t)N1), pd1 = pk.add(K1->desc, 32 * (size_t)N1), pu1 = pk.add(ur1.data(), sizeof(float) * ur1.size()),
                 pk2 = pk.add(kps2.data(), sizeof(KeyPointRec) * kps2.size()), pd2 = pk.add(desc2.data(), desc2.size()), pu2 = pk.add(ur2.data(), sizeof(float) * ur2.size()),
                 pm2 = pk.add(mp2.data(), mp2.size()), pit = pk.add(items.data(), sizeof(BowItem) * items.size()), pf2 = pk.add(feat2.data(), sizeof(int) * feat2.size()),
                 pps = pk.add(Ps.data(), sizeof(BowParams) * Ps.size());
    if (pk.flush() || h->d_si[SI_BEST].ensure(items.size())) return fail(ORBX_E_DEVICE, "upload/allocation failed");
    dim3 grid(((int)items.size() + 3) / 4, 1, 1), blk(256, 1, 1);
    if (kb8)
        ORBX_LAUNCH(k_bow_search_kb8, grid, blk, 0, h->s0, pk.dev<BowItem>(pit), (int)items.size(),
                    pk.dev<KeyPointRec>(pk1), pk.dev<unsigned long long>(pd1), pk.dev<float>(pu1),
                    pk.dev<KeyPointRec>(pk2), pk.dev<unsigned long long>(pd2), pk.dev<float>(pu2),
                    pk.dev<uint8_t>(pm2), pk.dev<int>(pf2), pk.dev<BowParams>(pps), h->d_si[SI_BEST].p);
    else
    ORBX_LAUNCH(k_bow_search, grid, blk, 0, h->s0, pk.dev<BowItem>(pit), (int)items.size(),
                pk.dev<KeyPointRec>(pk1), pk.dev<unsigned long long>(pd1), pk.dev<float>(pu1),
                pk.dev<KeyPointRec>(pk2), pk.dev<unsigned long long>(pd2), pk.dev<float>(pu2),
                pk.dev<uint8_t>(pm2), pk.dev<int>(pf2), pk.dev<BowParams>(pps), h->d_si[SI_BEST].p);
    std::vector<int> best(items.size());
    if (fetch_sync(h, best.data(), h->d_si[SI_BEST].p, sizeof(int) * items.size()) || rt::check_launch())
        return fail(ORBX_E_DEVICE, "bow search failed: %s", rt::last_error());
    for (int j = 0; j < n2; j++) {
        const OrbmKeyFrameView* K2 = K2s[j];
        int* m12 = matches12 + (size_t)j * N1;
        int nmatches = 0;
        std::vector<int> rotHist[HISTO_LENGTH];
        for (int k = first_item[j]; k < first_item[j + 1]; k++) {
            if (best[k] < 0) continue;
            const int idx1 = items[k].idx1, idx2 = best[k] - base2[j];
            m12[idx1] = idx2;
            nmatches++;
            if (check_ori) rotHist[rot_bin(K1->keys_un[idx1].angle, K2->keys_un[idx2].angle)].push_back(idx1);
        }
        if (check_ori) {
            int ind1 = -1, ind2 = -1, ind3 = -1;
            three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
            for (int i = 0; i < HISTO_LENGTH; i++) {
                if (i == ind1 || i == ind2 || i == ind3) continue;
                for (int idx1 : rotHist[i]) { m12[idx1] = -1; nmatches--; }
            }
        }
        if (nmatches_out) nmatches_out[j] = nmatches;
    }
    return ORBX_OK;
}

int orbm_search_for_triangulation_batch(orbx_extractor* h, const OrbmKeyFrameView* K1, int n2, const OrbmKeyFrameView* const* K2s, const float* F12s,
                                        const float* eps, int only_stereo, int coarse, int check_ori, int* matches12, int* nmatches_out) {
    if (n2 > 0 && !F12s) return fail(ORBX_E_ARG, "null");
    return search_for_triangulation_impl(h, K1, n2, K2s, F12s, eps, nullptr, only_stereo, coarse, check_ori, matches12, nmatches_out);
}

int orbm_search_for_triangulation_kb8(orbx_extractor* h, const OrbmKeyFrameView* K1, const OrbmKeyFrameView* K2, const OrbmKB8Pair* cams, const float ep[2],
                                      int only_stereo, int coarse, int check_ori, int* matches12, int* nmatches_out) {
    if (!h || !K1 || !K2 || !cams || !ep || !matches12) return fail(ORBX_E_ARG, "null");
    return search_for_triangulation_impl(h, K1, 1, &K2, nullptr, ep, cams, only_stereo, coarse, check_ori, matches12, nmatches_out);
}

int orbm_search_for_triangulation(orbx_extractor* h, const OrbmKeyFrameView* K1, const OrbmKeyFrameView* K2, const float F12[9], const float ep[2],
                                  int only_stereo, int coarse, int check_ori, int* matches12, int* nmatches_out) {
    if (!h || !K1 || !K2 || !F12 || !ep || !matches12) return fail(ORBX_E_ARG, "null");
    return orbm_search_for_triangulation_batch(h, K1, 1, &K2, F12, ep, only_stereo, coarse, check_ori, matches12, nmatches_out);
}


// ORBmatcher::SearchByBoW, both overloads (src/ORBmatcher.cc:259-493, :892-1043), non-fisheye path, for n independent (K1, K2) pairs in one
// launch: Tracking::Relocalization matches every candidate key frame against the current frame (src/Tracking.cc:4360-4380), loop / merge
// detection one key frame against its candidates' covisible key frames (src/LoopClosing.cc:840-850).  The pairs' descriptors and feature
// lists are concatenated (indices made global), one k_bow_dists launch computes every node-bucket distance, the order-dependent selection
// is replayed per pair on the host.
int orbm_search_by_bow_batch(orbx_extractor* h, int n, const OrbmKeyFrameView* const* K1s, const OrbmKeyFrameView* const* K2s, float nnratio, int th_inclusive,
                             int check_ori, int* const* matches12, int* nmatches_out) {
    if (!h || n < 0 || (n > 0 && (!K1s || !K2s || !matches12))) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    std::vector<BowItem> items;
    std::vector<int> first_item(n + 1, 0), base1(n + 1, 0), base2(n + 1, 0), fbase2(n + 1, 0), obase(n + 1, 0);
    int total = 0;
    for (int p = 0; p < n; p++) {
        const OrbmKeyFrameView *K1 = K1s[p], *K2 = K2s[p];
        if (!K1 || !K2 || !matches12[p] || K1->N >= 65535 || K2->N >= 65535) return fail(ORBX_E_ARG, "bad key frame pair %d", p);
        for (int i = 0; i < K1->N; i++) matches12[p][i] = -1;
        if (nmatches_out) nmatches_out[p] = 0;
        first_item[p] = (int)items.size(); obase[p] = total;
        base1[p + 1] = base1[p] + K1->N; base2[p + 1] = base2[p] + K2->N; fbase2[p + 1] = fbase2[p] + K2->fv_start[K2->fv_nodes];
        int a = 0, b = 0;
        while (a < K1->fv_nodes && b < K2->fv_nodes) {
            const uint32_t na = K1->fv_node_id[a], nb = K2->fv_node_id[b];
            if (na == nb) {
                for (int k = K1->fv_start[a]; k < K1->fv_start[a + 1]; k++) {
                    const int idx1 = (int)K1->fv_feat[k];
                    if (!K1->has_map_point || !K1->has_map_point[idx1]) continue;     // !pMP || pMP->isBad()
                    BowItem it; it.idx1 = base1[p] + idx1; it.start2 = fbase2[p] + K2->fv_start[b]; it.cnt2 = K2->fv_start[b + 1] - K2->fv_start[b]; it.out_off = total;
                    total += it.cnt2;
                    items.push_back(it);
                }
                a++; b++;
            } else if (na < nb) { while (a < K1->fv_nodes && K1->fv_node_id[a] < nb) a++; }
            else { while (b < K2->fv_nodes && K2->fv_node_id[b] < na) b++; }
        }
    }
    first_item[n] = (int)items.size(); obase[n] = total;
    if (items.empty() || total == 0) return ORBX_OK;
    const int T1 = std::max(base1[n], 1), T2 = std::max(base2[n], 1), TF = std::max(fbase2[n], 1);
    std::vector<uint8_t> desc1((size_t)T1 * 32), desc2((size_t)T2 * 32), elig(T2, 1);
    std::vector<int> feat2(TF);
    for (int p = 0; p < n; p++) {
        const OrbmKeyFrameView *K1 = K1s[p], *K2 = K2s[p];
        if (K1->N > 0) memcpy(&desc1[(size_t)base1[p] * 32], K1->desc, 32 * (size_t)K1->N);
        if (K2->N > 0) memcpy(&desc2[(size_t)base2[p] * 32], K2->desc, 32 * (size_t)K2->N);
        if (K2->has_map_point) memcpy(&elig[base2[p]], K2->has_map_point, K2->N);
        const int nf = K2->fv_start[K2->fv_nodes];
        for (int k = 0; k < nf; k++) feat2[fbase2[p] + k] = (int)K2->fv_feat[k] + base2[p];
    }
    Packer pk(h);
    const size_t pd1 = pk.add(desc1.data(), desc1.size()), pd2 = pk.add(desc2.data(), desc2.size()), pel = pk.add(elig.data(), elig.size()),
                 pit = pk.add(items.data(), sizeof(BowItem) * items.size()), pf2 = pk.add(feat2.data(), sizeof(int) * feat2.size());
    if (pk.flush() || h->d_si[SI_BEST].ensure((size_t)total + 1)) return fail(ORBX_E_DEVICE, "upload/allocation failed");
    dim3 grid(((int)items.size() + 3) / 4, 1, 1), blk(256, 1, 1);
    ORBX_LAUNCH(k_bow_dists, grid, blk, 0, h->s0, pk.dev<BowItem>(pit), (int)items.size(), pk.dev<unsigned long long>(pd1), pk.dev<unsigned long long>(pd2),
                pk.dev<uint8_t>(pel), pk.dev<int>(pf2), h->d_si[SI_BEST].p);
    std::vector<int> dist((size_t)total);
    if (fetch_sync(h, dist.data(), h->d_si[SI_BEST].p, sizeof(int) * (size_t)total) || rt::check_launch())
        return fail(ORBX_E_DEVICE, "bow distances failed: %s", rt::last_error());
    for (int p = 0; p < n; p++) {
        const OrbmKeyFrameView *K1 = K1s[p], *K2 = K2s[p];
        const int N2 = K2->N;
        int* m12 = matches12[p];
        int nmatches = 0;
        // sequential replay: a target taken by an earlier feature is skipped (:331 vpMapPointMatches[realIdxF], :948 vbMatched2[idx2])
        std::vector<uint8_t> taken(N2 > 0 ? N2 : 1, 0);
        std::vector<int> rotHist[HISTO_LENGTH];
        for (int q = first_item[p]; q < first_item[p + 1]; q++) {
            const BowItem& it = items[q];
            const int idx1 = it.idx1 - base1[p];
            int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
            for (int j = 0; j < it.cnt2; j++) {
                const int idx2 = feat2[it.start2 + j] - base2[p];
                const int d = dist[(size_t)it.out_off + j];
                if (taken[idx2] || d < 0) continue;
                if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdx2 = idx2; }
                else if (d < bestDist2) bestDist2 = d;
            }
            const bool pass = th_inclusive ? (bestDist1 <= TH_LOW) : (bestDist1 < TH_LOW);
            if (pass && static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                m12[idx1] = bestIdx2;
                taken[bestIdx2] = 1;
                nmatches++;
                if (check_ori) rotHist[rot_bin(K1->keys_un[idx1].angle, K2->keys_un[bestIdx2].angle)].push_back(idx1);
            }
        }
        if (check_ori) {
            int ind1 = -1, ind2 = -1, ind3 = -1;
            three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
            for (int i = 0; i < HISTO_LENGTH; i++) {
                if (i == ind1 || i == ind2 || i == ind3) continue;
                for (int idx1 : rotHist[i]) { m12[idx1] = -1; nmatches--; }
            }
        }
        if (nmatches_out) nmatches_out[p] = nmatches;
    }
    return ORBX_OK;
}

// ---- device-resident key frames ------------------------------------------------------------------------------------------------
// What the vocabulary-bucket searches read of a key frame (keys, descriptors, mvuRight, mFeatVec) is uploaded once; the searches then move
// only their call-time state (map point flags, poses) to the device and their result back.  The host keeps the angles for the
// rotation-consistency histograms and the per-level scales.
struct orbm_keyframe {
    int device = 0, N = 0, fv_nodes = 0, nlevels = 0;
    uint8_t* dmem = nullptr;
    ResidentKF dev;
    std::vector<float> angle;
    float scale[kMaxLevels], sigma2[kMaxLevels];
};

int orbm_keyframe_create(orbx_extractor* h, const OrbmKeyFrameView* K, orbm_keyframe** out) {
    if (!h || !K || !out || K->N < 0 || K->fv_nodes < 0 || (K->N > 0 && (!K->keys_un || !K->desc))) return fail(ORBX_E_ARG, "bad key frame view");
    if (K->N >= 65535 || K->nlevels > kMaxLevels) return fail(ORBX_E_ARG, "key frame too large");
    if (K->fv_nodes > 0 && (!K->fv_node_id || !K->fv_start || !K->fv_feat)) return fail(ORBX_E_ARG, "bad feature vector");
    rt::set_device(h->device);
    const int N = K->N, N1 = std::max(N, 1), nn = K->fv_nodes, nf = nn > 0 ? K->fv_start[nn] : 0;
    const size_t okp = 0, odesc = okp + al16(sizeof(KeyPointRec) * (size_t)N1), our = odesc + al16(32 * (size_t)N1), onid = our + al16(4 * (size_t)N1),
                 ost = onid + al16(4 * (size_t)(nn + 1)), oft = ost + al16(4 * (size_t)(nn + 1)), onof = oft + al16(4 * (size_t)(nf + 1)),
                 total = onof + al16(4 * (size_t)N1);
    std::vector<uint8_t> stage(total, 0);
    if (N > 0) { memcpy(&stage[okp], K->keys_un, sizeof(KeyPointRec) * (size_t)N); memcpy(&stage[odesc], K->desc, 32 * (size_t)N); }
    float* ur = (float*)&stage[our];
    for (int i = 0; i < N1; i++) ur[i] = (K->u_right && i < N) ? K->u_right[i] : -1.0f;
    int* nof = (int*)&stage[onof];
    for (int i = 0; i < N1; i++) nof[i] = -1;
    if (nn > 0) {
        memcpy(&stage[onid], K->fv_node_id, 4 * (size_t)nn); memcpy(&stage[ost], K->fv_start, 4 * (size_t)(nn + 1));
        int* ft = (int*)&stage[oft];
        for (int a = 0; a < nn; a++) {
            if (a > 0 && K->fv_node_id[a] <= K->fv_node_id[a - 1]) return fail(ORBX_E_ARG, "feature vector node ids not ascending");
            for (int k = K->fv_start[a]; k < K->fv_start[a + 1]; k++) {
                const int f = (int)K->fv_feat[k];
                if (f < 0 || f >= N) return fail(ORBX_E_ARG, "feature vector entry out of range");
                ft[k] = f; nof[f] = a;
            }
        }
    }
    orbm_keyframe* kf = new orbm_keyframe();
    kf->device = h->device; kf->N = N; kf->fv_nodes = nn; kf->nlevels = K->nlevels;
    kf->dmem = (uint8_t*)rt::dmalloc(total);
    if (!kf->dmem || rt::copy_h2d(kf->dmem, stage.data(), total, h->s0) || rt::stream_sync(h->s0)) { rt::dfree(kf->dmem); delete kf; return fail(ORBX_E_DEVICE, "key frame upload failed"); }
    kf->dev.kps = (const KeyPointRec*)(kf->dmem + okp); kf->dev.desc = (const unsigned long long*)(kf->dmem + odesc); kf->dev.ur = (const float*)(kf->dmem + our);
    kf->dev.node_id = (const uint32_t*)(kf->dmem + onid); kf->dev.fv_start = (const int*)(kf->dmem + ost); kf->dev.fv_feat = (const int*)(kf->dmem + oft);
    kf->dev.node_of_feat = (const int*)(kf->dmem + onof); kf->dev.N = N; kf->dev.fv_nodes = nn;
    kf->angle.resize(N1);
    for (int i = 0; i < N; i++) kf->angle[i] = K->keys_un[i].angle;
    for (int l = 0; l < kMaxLevels; l++) { kf->scale[l] = l < K->nlevels && K->scale_factors ? K->scale_factors[l] : 1.0f; kf->sigma2[l] = l < K->nlevels && K->level_sigma2 ? K->level_sigma2[l] : 1.0f; }
    *out = kf;
    return ORBX_OK;
}

void orbm_keyframe_destroy(orbm_keyframe* kf) {
    if (!kf) return;
    rt::set_device(kf->device);
    rt::dfree(kf->dmem);
    delete kf;
}

// rotation-consistency pruning shared by the resident searches: m12 entries outside the three strongest bins of the angle-difference
// histogram are reset (src/ORBmatcher.cc:1270-1286); returns the number of matches left
static int prune_by_rotation(const orbm_keyframe* K1, const orbm_keyframe* K2, int* m12, bool check_ori) {
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < K1->N; i++) {
        if (m12[i] < 0) continue;
        nmatches++;
        if (check_ori) rotHist[rot_bin(K1->angle[i], K2->angle[m12[i]])].push_back(i);
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i]) { m12[idx1] = -1; nmatches--; }
        }
    }
    return nmatches;
}

static int sft_resident_impl(orbx_extractor* h, orbm_keyframe* K1, const uint8_t* has_mp1, int n2, orbm_keyframe* const* K2s,
                             const uint8_t* const* has_mp2, const float* F12s, const OrbmKB8Pair* kb8, const float* eps, int only_stereo, int coarse, int check_ori,
                             int* matches12, int* nmatches_out) {
    if (!h || !K1 || n2 < 0 || (n2 > 0 && (!K2s || (!F12s && !kb8) || !eps || !matches12))) return fail(ORBX_E_ARG, "null");
    if (K1->device != h->device) return fail(ORBX_E_ARG, "key frame lives on another device");
    rt::set_device(h->device);
    const int N1 = K1->N;
    if (n2 == 0 || N1 == 0) { for (int j = 0; j < n2; j++) if (nmatches_out) nmatches_out[j] = 0; return ORBX_OK; }
    // call-time flags: KF1's map point flags at 0, then each neighbour's
    size_t ftotal = al16((size_t)N1);
    std::vector<SftNeighbour> nb(n2);
    for (int j = 0; j < n2; j++) {
        const orbm_keyframe* K2 = K2s[j];
        if (!K2 || K2->device != h->device) return fail(ORBX_E_ARG, "bad neighbour key frame %d", j);
        SftNeighbour& S = nb[j]; memset(&S, 0, sizeof S);
        S.k2 = K2->dev; S.mp2_off = (int)ftotal; ftotal += al16((size_t)std::max(K2->N, 1));
        BowParams& P = S.P;
        if (F12s) for (int i = 0; i < 9; i++) P.F12[i] = F12s[9 * (size_t)j + i];
        P.ep[0] = eps[2 * (size_t)j]; P.ep[1] = eps[2 * (size_t)j + 1];
        for (int l = 0; l < kMaxLevels; l++) { P.scale2[l] = K2->scale[l]; P.sigma2_2[l] = K2->sigma2[l]; }
        P.only_stereo = only_stereo; P.coarse = coarse; P.th_low = TH_LOW; P.nleft1 = P.nleft2 = -1;
        if (kb8) {                               // Kannala-Brandt cameras: one OrbmKB8Pair per neighbour (as orbm_search_for_triangulation_kb8)
            const OrbmKB8Pair& C = kb8[j];
            P.kb8 = 1; P.nleft1 = C.nleft1; P.nleft2 = C.nleft2;
            memcpy(P.cam1, C.cam1, sizeof P.cam1); memcpy(P.cam2, C.cam2, sizeof P.cam2); memcpy(P.R, C.R, sizeof P.R); memcpy(P.t, C.t, sizeof P.t);
            for (int l = 0; l < kMaxLevels; l++) P.sigma2_1[l] = K1->sigma2[l];
        }
    }
    std::vector<uint8_t> flags(ftotal, 0);
    if (has_mp1) memcpy(flags.data(), has_mp1, (size_t)N1);
    for (int j = 0; j < n2; j++) if (has_mp2 && has_mp2[j] && K2s[j]->N > 0) memcpy(&flags[nb[j].mp2_off], has_mp2[j], (size_t)K2s[j]->N);
    Packer pk(h);
    const size_t pf = pk.add(flags.data(), flags.size()), pn = pk.add(nb.data(), sizeof(SftNeighbour) * nb.size());
    if (pk.flush() || h->d_si[SI_BEST].ensure((size_t)n2 * N1)) return fail(ORBX_E_DEVICE, "upload/allocation failed");
    dim3 grid((N1 + 3) / 4, n2, 1), blk(256, 1, 1);
    if (kb8) ORBX_LAUNCH(k_sft_resident_kb8, grid, blk, 0, h->s0, K1->dev, pk.dev<uint8_t>(pf), pk.dev<SftNeighbour>(pn), h->d_si[SI_BEST].p);
    else ORBX_LAUNCH(k_sft_resident, grid, blk, 0, h->s0, K1->dev, pk.dev<uint8_t>(pf), pk.dev<SftNeighbour>(pn), h->d_si[SI_BEST].p);
    if (fetch_sync(h, matches12, h->d_si[SI_BEST].p, sizeof(int) * (size_t)n2 * N1) || rt::check_launch())
        return fail(ORBX_E_DEVICE, "resident triangulation search failed: %s", rt::last_error());
    for (int j = 0; j < n2; j++) {
        const int nm = prune_by_rotation(K1, K2s[j], matches12 + (size_t)j * N1, check_ori != 0);
        if (nmatches_out) nmatches_out[j] = nm;
    }
    return ORBX_OK;
}

int orbm_search_for_triangulation_resident(orbx_extractor* h, orbm_keyframe* K1, const uint8_t* has_mp1, int n2, orbm_keyframe* const* K2s,
                                           const uint8_t* const* has_mp2, const float* F12s, const float* eps, int only_stereo, int coarse, int check_ori,
                                           int* matches12, int* nmatches_out) {
    if (n2 > 0 && !F12s) return fail(ORBX_E_ARG, "null");
    return sft_resident_impl(h, K1, has_mp1, n2, K2s, has_mp2, F12s, nullptr, eps, only_stereo, coarse, check_ori, matches12, nmatches_out);
}

int orbm_search_for_triangulation_resident_kb8(orbx_extractor* h, orbm_keyframe* K1, const uint8_t* has_mp1, int n2, orbm_keyframe* const* K2s,
                                               const uint8_t* const* has_mp2, const OrbmKB8Pair* cams, const float* eps, int only_stereo, int coarse, int check_ori,
                                               int* matches12, int* nmatches_out) {
    if (n2 > 0 && !cams) return fail(ORBX_E_ARG, "null");
    return sft_resident_impl(h, K1, has_mp1, n2, K2s, has_mp2, nullptr, cams, eps, only_stereo, coarse, check_ori, matches12, nmatches_out);
}

int orbm_search_by_bow_resident(orbx_extractor* h, int n, orbm_keyframe* const* K1s, const uint8_t* const* has_mp1, orbm_keyframe* const* K2s,
                                const uint8_t* const* eligible2, float nnratio, int th_inclusive, int check_ori, int* const* matches12, int* nmatches_out) {
    if (!h || n < 0 || (n > 0 && (!K1s || !K2s || !matches12))) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    if (n == 0) return ORBX_OK;
    size_t ftotal = 0; int N1cap = 1, maxnodes = 1;
    std::vector<BowPairResident> pairs(n);
    for (int p = 0; p < n; p++) {
        const orbm_keyframe *K1 = K1s[p], *K2 = K2s[p];
        if (!K1 || !K2 || !matches12[p] || K1->device != h->device || K2->device != h->device) return fail(ORBX_E_ARG, "bad key frame pair %d", p);
        BowPairResident& B = pairs[p];
        B.k1 = K1->dev; B.k2 = K2->dev; B.k2_nodes_dev = nullptr;
        B.mp1_off = -1; B.elig2_off = -1;
        if (has_mp1 && has_mp1[p] && K1->N > 0) { B.mp1_off = (int)ftotal; ftotal += al16((size_t)K1->N); }
        if (eligible2 && eligible2[p] && K2->N > 0) { B.elig2_off = (int)ftotal; ftotal += al16((size_t)K2->N); }
        N1cap = std::max(N1cap, K1->N); maxnodes = std::max(maxnodes, K1->fv_nodes);
    }
    std::vector<uint8_t> flags(std::max<size_t>(ftotal, 16), 0);
    for (int p = 0; p < n; p++) {
        if (pairs[p].mp1_off >= 0) memcpy(&flags[pairs[p].mp1_off], has_mp1[p], (size_t)K1s[p]->N);
        if (pairs[p].elig2_off >= 0) memcpy(&flags[pairs[p].elig2_off], eligible2[p], (size_t)K2s[p]->N);
    }
    Packer pk(h);
    const size_t pf = pk.add(flags.data(), flags.size()), pp = pk.add(pairs.data(), sizeof(BowPairResident) * pairs.size());
    const size_t nout = (size_t)n * N1cap;
    if (pk.flush() || h->d_si[SI_BEST].ensure(nout + 4)) return fail(ORBX_E_DEVICE, "upload/allocation failed");
    // results pre-set to -1, the status word behind them to 0
    if (rt::memset_async(h->d_si[SI_BEST].p, 0xFF, sizeof(int) * nout, h->s0) || rt::memset_async(h->d_si[SI_BEST].p + nout, 0, sizeof(int) * 4, h->s0))
        return fail(ORBX_E_DEVICE, "memset failed");
    dim3 grid((maxnodes + 3) / 4, n, 1), blk(256, 1, 1);
    ORBX_LAUNCH(k_bow_match_resident, grid, blk, 0, h->s0, pk.dev<BowPairResident>(pp), pk.dev<uint8_t>(pf), nnratio, TH_LOW, th_inclusive,
                h->d_si[SI_BEST].p, N1cap, h->d_si[SI_BEST].p + nout);
    std::vector<int> res(nout + 4);
    if (fetch_sync(h, res.data(), h->d_si[SI_BEST].p, sizeof(int) * (nout + 4)) || rt::check_launch())
        return fail(ORBX_E_DEVICE, "resident bow search failed: %s", rt::last_error());
    if (res[nout] & 4) return fail(ORBX_E_CAPACITY, "a vocabulary node holds more than 2048 features of one key frame");
    for (int p = 0; p < n; p++) {
        const int N1 = K1s[p]->N;
        if (N1 > 0) memcpy(matches12[p], &res[(size_t)p * N1cap], sizeof(int) * (size_t)N1);
        const int nm = prune_by_rotation(K1s[p], K2s[p], matches12[p], check_ori != 0);
        if (nmatches_out) nmatches_out[p] = nm;
    }
    return ORBX_OK;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vpMapPointMatches) (src/ORBmatcher.cc:259-493) for the frames of a batch: frame b of the last
// extraction (its FeatureVector where orbv_transform_extracted left it) against the resident key frame KFs[b].  Two launches for the whole batch:
// k_bow_match_resident (the accept loop per vocabulary node) and k_bow_rotation_prune (histogram, three maxima, resets, counts) - nothing but the
// call-time flags goes up, nothing but the matches comes down.
int orbm_search_by_bow_frames_batch(orbx_extractor* h, const orbv_vocabulary* v, int first, int B, orbm_keyframe* const* KFs, const uint8_t* const* has_mp1,
                                    float nnratio, int check_ori, int* const* matches12, int* nmatches_out) {
    if (!h || !v || B <= 0 || !KFs || !has_mp1 || !matches12) return fail(ORBX_E_ARG, "null");
    VocFrameArrays V;
    if (orbv_frame_arrays(v, &V) || V.handle != (const void*)h || V.first != first || V.lastB != B || V.cap != h->kp_total_cap || first < 0 || first + B > h->lastB)
        return fail(ORBX_E_ARG, "run orbv_transform_extracted(v, h, %d, %d, levelsup) on this extraction first: the FeatureVectors of these frames are read where it leaves them", first, B);
    if (V.extract_gen != h->extract_gen)
        return fail(ORBX_E_ARG, "the handle has extracted another batch since orbv_transform_extracted ran: its FeatureVectors index the keypoints of the previous batch - transform again");
    if (V.device != h->device) return fail(ORBX_E_ARG, "vocabulary and extractor live on different devices");
    if (undistort_stale(h)) return fail(ORBX_E_ARG, "orbx_set_undistort was called after the last extraction: extract again before searching its frames");
    rt::set_device(h->device);
    const int cap = h->kp_total_cap;
    size_t ftotal = 0; int N1cap = 1, maxnodes = 1;
    std::vector<BowPairResident> pairs(B);
    for (int b = 0; b < B; b++) {
        const orbm_keyframe* K1 = KFs[b];
        if (!K1 || !matches12[b] || K1->device != h->device) return fail(ORBX_E_ARG, "bad key frame of frame %d", b);
        BowPairResident& P = pairs[b];
        memset(&P, 0, sizeof P);
        P.k1 = K1->dev;
        P.k2.kps = (h->ex_undist_active ? h->d_kps_un.p : h->d_kps.p) + (size_t)(first + b) * cap;          // F.mvKeysUn (the angle is what is read of it)
        P.k2.desc = h->d_desc.p + (size_t)(first + b) * cap * 4; P.k2.ur = nullptr;
        P.k2.node_id = V.fv_node + (size_t)b * cap; P.k2.fv_start = V.fv_start + (size_t)b * (cap + 1); P.k2.fv_feat = V.fv_feat + (size_t)b * cap;
        P.k2.node_of_feat = nullptr; P.k2.N = cap; P.k2.fv_nodes = 0;
        P.k2_nodes_dev = V.nout + 2 * b + 1;
        P.mp1_off = -1; P.elig2_off = -1;
        if (has_mp1[b] && K1->N > 0) { P.mp1_off = (int)ftotal; ftotal += al16((size_t)K1->N); }
        N1cap = std::max(N1cap, K1->N); maxnodes = std::max(maxnodes, K1->fv_nodes);
    }
    std::vector<uint8_t> flags(std::max<size_t>(ftotal, 16), 0);
    for (int b = 0; b < B; b++) if (pairs[b].mp1_off >= 0) memcpy(&flags[pairs[b].mp1_off], has_mp1[b], (size_t)KFs[b]->N);
    Packer pk(h);
    const size_t pf = pk.add(flags.data(), flags.size()), pp = pk.add(pairs.data(), sizeof(BowPairResident) * pairs.size());
    const size_t nout = (size_t)B * N1cap, ntot = nout + 4 + (size_t)B;            // matches | status | nmatches
    if (pk.flush() || h->d_si[SI_BEST].ensure(ntot)) return fail(ORBX_E_DEVICE, "upload/allocation failed");
    if (rt::memset_async(h->d_si[SI_BEST].p, 0xFF, sizeof(int) * nout, h->s0) || rt::memset_async(h->d_si[SI_BEST].p + nout, 0, sizeof(int) * (4 + (size_t)B), h->s0))
        return fail(ORBX_E_DEVICE, "memset failed");
    if (h->profile) rt::event_record(h->ev_stage[ST_MATCH][0], h->s0);
    dim3 grid((maxnodes + 3) / 4, B, 1), blk(256, 1, 1);
    ORBX_LAUNCH(k_bow_match_resident, grid, blk, 0, h->s0, pk.dev<BowPairResident>(pp), pk.dev<uint8_t>(pf), nnratio, TH_LOW, 1, h->d_si[SI_BEST].p, N1cap,
                h->d_si[SI_BEST].p + nout);
    dim3 gridp(B, 1, 1), blkp(64, 1, 1);
    ORBX_LAUNCH(k_bow_rotation_prune, gridp, blkp, 0, h->s0, pk.dev<BowPairResident>(pp), h->d_si[SI_BEST].p, N1cap, check_ori, h->d_si[SI_BEST].p + nout + 4);
    if (h->profile) rt::event_record(h->ev_stage[ST_MATCH][1], h->s0);
    std::vector<int> res(ntot);
    if (fetch_sync(h, res.data(), h->d_si[SI_BEST].p, sizeof(int) * ntot) || rt::check_launch())
        return fail(ORBX_E_DEVICE, "batched bow search failed: %s", rt::last_error());
    if (res[nout] & 4) return fail(ORBX_E_CAPACITY, "a vocabulary node holds more than 2048 features of one frame");
    for (int b = 0; b < B; b++) {
        const int N1 = KFs[b]->N;
        if (N1 > 0) memcpy(matches12[b], &res[(size_t)b * N1cap], sizeof(int) * (size_t)N1);
        if (nmatches_out) nmatches_out[b] = res[nout + 4 + b];
    }
    return ORBX_OK;
}

int orbm_search_by_bow(orbx_extractor* h, const OrbmKeyFrameView* K1, const OrbmKeyFrameView* K2, float nnratio, int th_inclusive,
                       int check_ori, int* matches12, int* nmatches_out) {
    if (!h || !K1 || !K2 || !matches12) return fail(ORBX_E_ARG, "null");
    int nm = 0;
    const int rc = orbm_search_by_bow_batch(h, 1, &K1, &K2, nnratio, th_inclusive, check_ori, &matches12, &nm);
    if (nmatches_out) *nmatches_out = nm;
    return rc;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) for a fisheye-rig frame (F.Nleft != -1, src/ORBmatcher.cc:259-493 incl.
// :343-372 and :414-446): best / second best are kept separately for the features of camera 1 (index < nleft2) and of camera 2, the
// camera-2 match is only considered when the camera-1 best passed TH_LOW (:378, :414 nest that way) and takes no ratio test (":417 || true").
// assigned2[j] = feature of K1 whose map point goes to vpMapPointMatches[j], or -1.
int orbm_search_by_bow_fisheye(orbx_extractor* h, const OrbmKeyFrameView* K1, const OrbmKeyFrameView* K2, int nleft2, float nnratio, int check_ori,
                               int* assigned2, int* nmatches_out) {
    if (!h || !K1 || !K2 || !assigned2) return fail(ORBX_E_ARG, "null");
    if (K1->N >= 65535 || K2->N >= 65535 || nleft2 < 0 || nleft2 > K2->N) return fail(ORBX_E_ARG, "bad key frame / frame sizes");
    rt::set_device(h->device);
    const int N1 = K1->N, N2 = K2->N;
    for (int i = 0; i < N2; i++) assigned2[i] = -1;
    std::vector<BowItem> items;
    int a = 0, b = 0, total = 0;
    while (a < K1->fv_nodes && b < K2->fv_nodes) {
        const uint32_t na = K1->fv_node_id[a], nb = K2->fv_node_id[b];
        if (na == nb) {
            for (int k = K1->fv_start[a]; k < K1->fv_start[a + 1]; k++) {
                const int idx1 = (int)K1->fv_feat[k];
                if (!K1->has_map_point || !K1->has_map_point[idx1]) continue;     // !pMP || pMP->isBad()
                BowItem it; it.idx1 = idx1; it.start2 = K2->fv_start[b]; it.cnt2 = K2->fv_start[b + 1] - K2->fv_start[b]; it.out_off = total;
                total += it.cnt2;
                items.push_back(it);
            }
            a++; b++;
        } else if (na < nb) { while (a < K1->fv_nodes && K1->fv_node_id[a] < nb) a++; }
        else { while (b < K2->fv_nodes && K2->fv_node_id[b] < na) b++; }
    }
    int nmatches = 0;
    if (!items.empty() && total > 0) {
        std::vector<uint8_t> elig(N2 > 0 ? N2 : 1, 1);
        const int nfeat2 = K2->fv_start[K2->fv_nodes];
        Packer pk(h);
        const size_t pd1 = pk.add(K1->desc, 32 * (size_t)N1), pd2 = pk.add(K2->desc, 32 * (size_t)N2), pel = pk.add(elig.data(), elig.size()),
                     pit = pk.add(items.data(), sizeof(BowItem) * items.size()), pf2 = pk.add(K2->fv_feat, sizeof(int) * (size_t)nfeat2);
        if (pk.flush() || h->d_si[SI_BEST].ensure((size_t)total + 1)) return fail(ORBX_E_DEVICE, "upload/allocation failed");
        dim3 grid(((int)items.size() + 3) / 4, 1, 1), blk(256, 1, 1);
        ORBX_LAUNCH(k_bow_dists, grid, blk, 0, h->s0, pk.dev<BowItem>(pit), (int)items.size(), pk.dev<unsigned long long>(pd1), pk.dev<unsigned long long>(pd2),
                    pk.dev<uint8_t>(pel), pk.dev<int>(pf2), h->d_si[SI_BEST].p);
        std::vector<int> dist((size_t)total);
        if (fetch_sync(h, dist.data(), h->d_si[SI_BEST].p, sizeof(int) * (size_t)total) || rt::check_launch())
            return fail(ORBX_E_DEVICE, "bow distances failed: %s", rt::last_error());
        std::vector<int> rotHist[HISTO_LENGTH];
        for (const BowItem& it : items) {
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256, bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
            for (int j = 0; j < it.cnt2; j++) {
                const int idx2 = (int)K2->fv_feat[it.start2 + j];
                const int d = dist[(size_t)it.out_off + j];
                if (assigned2[idx2] >= 0) continue;                               // vpMapPointMatches[realIdxF] already set (:346)
                if (idx2 < nleft2) {
                    if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdxF = idx2; }
                    else if (d < bestDist2) bestDist2 = d;
                } else {
                    if (d < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = d; bestIdxFR = idx2; }
                    else if (d < bestDist2R) bestDist2R = d;
                }
            }
            if (bestDist1 <= TH_LOW) {
                if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                    assigned2[bestIdxF] = it.idx1;
                    nmatches++;
                    if (check_ori) rotHist[rot_bin(K1->keys_un[it.idx1].angle, K2->keys_un[bestIdxF].angle)].push_back(bestIdxF);
                }
                if (bestDist1R <= TH_LOW) {
                    assigned2[bestIdxFR] = it.idx1;
                    nmatches++;
                    if (check_ori) rotHist[rot_bin(K1->keys_un[it.idx1].angle, K2->keys_un[bestIdxFR].angle)].push_back(bestIdxFR);
                }
            }
        }
        if (check_ori) {
            int ind1 = -1, ind2 = -1, ind3 = -1;
            three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
            for (int i = 0; i < HISTO_LENGTH; i++) {
                if (i == ind1 || i == ind2 || i == ind3) continue;
                for (int idx2 : rotHist[i]) { assigned2[idx2] = -1; nmatches--; }
            }
        }
    }
    if (nmatches_out) *nmatches_out = nmatches;
    return ORBX_OK;
}

// ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:734-880)
int orbm_search_for_initialization(orbx_extractor* h, const OrbmFrameView* F1, const OrbmFrameView* F2, float* prev, int window_size,
                                   float nnratio, int check_ori, int* matches12, int* nmatches_out) {
    if (!h || !F1 || !F2 || !prev || !matches12) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    DeviceFrame D;
    int rc = upload_frame(h, F2, &D); if (rc) return rc;
    const int N1 = F1->N, N2 = F2->N;
    std::vector<AreaQuery> qs(N1);
    for (int i = 0; i < N1; i++) {
        AreaQuery& q = qs[i]; memset(&q, 0, sizeof q);
        const int level1 = F1->keys_un[i].octave;
        if (level1 > 0) continue;                                   // :755-757 only level-0 keypoints
        q.x = prev[2 * i]; q.y = prev[2 * i + 1]; q.r = (float)window_size; q.min_level = level1; q.max_level = level1; q.active = 1; q.gate = 0;
    }
    Csr c;
    rc = run_area_search(h, D, qs, F1->desc, &c); if (rc) return rc;
    for (int i = 0; i < N1; i++) matches12[i] = -1;
    std::vector<int> matched_dist(N2 > 0 ? N2 : 1, 0x7FFFFFFF), matches21(N2 > 0 ? N2 : 1, -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        if (!qs[i1].active || c.count[i1] == 0) continue;
        int bestDist = 0x7FFFFFFF, bestDist2 = 0x7FFFFFFF, bestIdx2 = -1;
        for (int k = 0; k < c.count[i1]; k++) {
            const int i2 = c.ent[2 * (size_t)(c.start[i1] + k)], dist = c.ent[2 * (size_t)(c.start[i1] + k) + 1] & 0xFFFF;
            if (matched_dist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2; matches21[bestIdx2] = i1; matched_dist[bestIdx2] = bestDist;
                nmatches++;
                if (check_ori) rotHist[rot_bin(F1->keys_un[i1].angle, F2->keys_un[bestIdx2].angle)].push_back(i1);
            }
        }
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i]) if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < N1; i1++) if (matches12[i1] >= 0) { prev[2 * i1] = F2->keys_un[matches12[i1]].x; prev[2 * i1 + 1] = F2->keys_un[matches12[i1]].y; }
    if (nmatches_out) *nmatches_out = nmatches;
    return ORBX_OK;
}

}  // extern "C"

// ---- the remaining projection-type searches.  The caller evaluates the geometry (Sophus/Eigen op order
// stays the reference's) and hands over, per map point: valid (= survived every skip test in front of GetFeaturesInArea), the
// projection, the predicted level and the descriptor.  GetFeaturesInArea, the level window, the chi-square gate and all Hamming
// distances run on the GPU (k_area_search); the accept logic is replayed here over the ordered candidate lists.
namespace {

// one k_area_search over P's valid points: window radius th * scale[pred], levels [pred-1, pred+hi]
int projected_candidates(orbx_extractor* h, const OrbmFrameView* F, const OrbmProjectedPointView* P, float th, int hi, int gate,
                         const float* inv_sigma2, std::vector<AreaQuery>* qs, Csr* c) {
    if (!P || P->M < 0 || (P->M > 0 && (!P->valid || !P->u || !P->v || !P->pred_level || !P->desc))) return fail(ORBX_E_ARG, "bad point view");
    if (F && F->nlevels > kMaxLevels) return fail(ORBX_E_ARG, "too many levels");
    DeviceFrame D;
    int rc = upload_frame(h, F, &D); if (rc) return rc;
    if (gate == 2) {
        if (!inv_sigma2) return fail(ORBX_E_ARG, "chi-square gate without mvInvLevelSigma2");
        for (int l = 0; l < F->nlevels; l++) D.g.inv_sigma2[l] = inv_sigma2[l];
    }
    const int M = P->M;
    qs->resize(M);
    for (int i = 0; i < M; i++) {
        AreaQuery& q = (*qs)[i]; memset(&q, 0, sizeof q);
        if (!P->valid[i]) continue;
        const int lvl = P->pred_level[i];
        if (lvl < 0 || lvl >= F->nlevels) continue;
        q.x = P->u[i]; q.y = P->v[i]; q.r = th * F->scale_factors[lvl];
        q.ur = P->ur ? P->ur[i] : 0.0f;
        q.min_level = lvl - 1; q.max_level = lvl + hi; q.active = 1; q.gate = gate;
    }
    return run_area_search(h, D, *qs, P->desc, c);
}

}  // namespace

extern "C" {

int orbm_project_points(orbx_extractor* h, const OrbmProjection* S, const OrbmProjectIn* in, const OrbmProjectOut* out) {
    if (!h || !S || !in || !out || in->M < 0) return fail(ORBX_E_ARG, "null");
    const int M = in->M;
    if (M == 0) return ORBX_OK;
    if (!in->pos || (S->angle_test && !in->normal) || (S->distance_test && (!in->min_inv || !in->max_inv))) return fail(ORBX_E_ARG, "projection inputs missing");
    static_assert(sizeof(ProjectParams) == sizeof(OrbmProjection), "OrbmProjection and ProjectParams describe the same record");
    rt::set_device(h->device);
    const size_t M1 = M;
    const size_t op = 0, on = op + al16(12 * M1), omn = on + al16(12 * M1), omx = omn + al16(4 * M1), os = omx + al16(4 * M1), in_total = os + al16(M1);
    const size_t ov = 0, oo = al16(M1), out_total = oo + al16(20 * M1);
    if (h->h_packB.ensure(in_total + 16) || h->d_sr[SR_QUERY].ensure(in_total + 16) || h->d_sr[SR_SPARE].ensure(out_total + 16) || h->h_out.ensure(out_total + 16))
        return fail(ORBX_E_DEVICE, "allocation failed");
    uint8_t* hp = h->h_packB.p;
    memcpy(hp + op, in->pos, 12 * M1);
    if (in->normal) memcpy(hp + on, in->normal, 12 * M1);
    if (in->min_inv) memcpy(hp + omn, in->min_inv, 4 * M1);
    if (in->max_inv) memcpy(hp + omx, in->max_inv, 4 * M1);
    if (in->skip) memcpy(hp + os, in->skip, M1); else memset(hp + os, 0, M1);
    if (rt::copy_h2d(h->d_sr[SR_QUERY].p, hp, in_total, h->s0)) return fail(ORBX_E_DEVICE, "upload failed");
    ProjectParams P; memcpy(&P, S, sizeof P);
    const uint8_t* di = h->d_sr[SR_QUERY].p; uint8_t* dout = h->d_sr[SR_SPARE].p;
    dim3 grid((M + 255) / 256, 1, 1), blk(256, 1, 1);
    ORBX_LAUNCH(k_project_points, grid, blk, 0, h->s0, P, M, (const float*)(di + op), (const float*)(di + on), (const float*)(di + omn), (const float*)(di + omx),
                (const uint8_t*)(di + os), dout + ov, (float*)(dout + oo), h->debug_stereo_flags);
    if (rt::copy_d2h(h->h_out.p, dout, out_total, h->s0) || rt::stream_sync(h->s0) || rt::check_launch()) return fail(ORBX_E_DEVICE, "projection kernel failed: %s", rt::last_error());
    const float* f = (const float*)(h->h_out.p + oo);
    if (out->valid) memcpy(out->valid, h->h_out.p + ov, M1);
    if (out->u) memcpy(out->u, f, 4 * M1);
    if (out->v) memcpy(out->v, f + M1, 4 * M1);
    if (out->ur) memcpy(out->ur, f + 2 * M1, 4 * M1);
    if (out->inv_z) memcpy(out->inv_z, f + 3 * M1, 4 * M1);
    if (out->dist) memcpy(out->dist, f + 4 * M1, 4 * M1);
    return ORBX_OK;
}

int orbm_search_by_projection_sim3(orbx_extractor* h, const OrbmFrameView* KF, const OrbmProjectedPointView* P, float th, float ratio_hamming,
                                   int* assigned, int* nmatches_out) {
    if (!h || !KF || !P || !assigned) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    std::vector<AreaQuery> qs; Csr c;
    int rc = projected_candidates(h, KF, P, th, 0, 0, nullptr, &qs, &c); if (rc) return rc;
    const int N = KF->N;
    std::vector<uint8_t> occ(N > 0 ? N : 1, 0);
    if (KF->occupied) memcpy(occ.data(), KF->occupied, N);          // vpMatched[idx] != NULL
    for (int i = 0; i < N; i++) assigned[i] = -1;
    int nmatches = 0;
    for (int i = 0; i < P->M; i++) {                                 // src/ORBmatcher.cc:519-606 / :640-727
        if (!qs[i].active || c.count[i] == 0) continue;
        int bestDist = 256, bestIdx = -1;
        for (int k = 0; k < c.count[i]; k++) {
            const int idx = c.ent[2 * (size_t)(c.start[i] + k)], dist = c.ent[2 * (size_t)(c.start[i] + k) + 1] & 0xFFFF;
            if (occ[idx]) continue;
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestIdx >= 0 && bestDist <= TH_LOW * ratio_hamming) { assigned[bestIdx] = i; occ[bestIdx] = 1; nmatches++; }
    }
    if (nmatches_out) *nmatches_out = nmatches;
    return ORBX_OK;
}

int orbm_search_by_projection_keyframe(orbx_extractor* h, const OrbmFrameView* Cur, const OrbmProjectedPointView* P, float th, int orb_dist,
                                       int check_ori, int* assigned, int* nmatches_out) {
    if (!h || !Cur || !P || !assigned) return fail(ORBX_E_ARG, "null");
    if (check_ori && P->M > 0 && !P->angle) return fail(ORBX_E_ARG, "orientation check without key-frame angles");
    rt::set_device(h->device);
    std::vector<AreaQuery> qs; Csr c;
    int rc = projected_candidates(h, Cur, P, th, 1, 0, nullptr, &qs, &c); if (rc) return rc;
    const int N = Cur->N;
    std::vector<uint8_t> occ(N > 0 ? N : 1, 0);
    if (Cur->occupied) memcpy(occ.data(), Cur->occupied, N);        // CurrentFrame.mvpMapPoints[i2] != NULL
    for (int i = 0; i < N; i++) assigned[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    for (int i = 0; i < P->M; i++) {                                 // src/ORBmatcher.cc:2213-2290
        if (!qs[i].active || c.count[i] == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < c.count[i]; k++) {
            const int i2 = c.ent[2 * (size_t)(c.start[i] + k)], dist = c.ent[2 * (size_t)(c.start[i] + k) + 1] & 0xFFFF;
            if (occ[i2]) continue;
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestIdx2 >= 0 && bestDist <= orb_dist) {
            assigned[bestIdx2] = i; occ[bestIdx2] = 1; nmatches++;
            if (check_ori) rotHist[rot_bin(P->angle[i], Cur->keys_un[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { assigned[idx] = -2; nmatches--; }
    }
    if (nmatches_out) *nmatches_out = nmatches;
    return ORBX_OK;
}

int orbm_fuse_candidates(orbx_extractor* h, const OrbmFrameView* KF, const OrbmProjectedPointView* P, float th, int chi2_gate,
                         const float* inv_level_sigma2, int* best_idx, int* best_dist) {
    if (!h || !KF || !P || !best_idx) return fail(ORBX_E_ARG, "null");
    if (chi2_gate && P->M > 0 && !P->ur) return fail(ORBX_E_ARG, "chi-square gate without projected right coordinates");
    rt::set_device(h->device);
    std::vector<AreaQuery> qs; Csr c;
    int rc = projected_candidates(h, KF, P, th, 0, chi2_gate ? 2 : 0, inv_level_sigma2, &qs, &c); if (rc) return rc;
    for (int i = 0; i < P->M; i++) {                                 // src/ORBmatcher.cc:1421-1490 / :1601-1640
        int bestDist = 256, bestIdx = -1;
        if (qs[i].active)
            for (int k = 0; k < c.count[i]; k++) {
                const int idx = c.ent[2 * (size_t)(c.start[i] + k)], dist = c.ent[2 * (size_t)(c.start[i] + k) + 1] & 0xFFFF;
                if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
            }
        if (bestDist > TH_LOW) bestIdx = -1;                         // :1493 / :1643
        best_idx[i] = bestIdx;
        if (best_dist) best_dist[i] = bestIdx >= 0 ? bestDist : -1;
    }
    return ORBX_OK;
}

int orbm_search_by_sim3(orbx_extractor* h, const OrbmFrameView* KF1, const OrbmFrameView* KF2, const OrbmProjectedPointView* P1in2,
                        const OrbmProjectedPointView* P2in1, float th, int* matches12, int* nfound_out) {
    if (!h || !KF1 || !KF2 || !P1in2 || !P2in1 || !matches12) return fail(ORBX_E_ARG, "null");
    if (P1in2->M != KF1->N || P2in1->M != KF2->N) return fail(ORBX_E_ARG, "one projected point per key-frame feature expected");
    rt::set_device(h->device);
    const int N1 = KF1->N, N2 = KF2->N;
    std::vector<int> vnMatch1(N1 > 0 ? N1 : 1, -1), vnMatch2(N2 > 0 ? N2 : 1, -1);
    for (int dir = 0; dir < 2; dir++) {                              // src/ORBmatcher.cc:1735-1826 and :1829-1915
        const OrbmProjectedPointView* P = dir == 0 ? P1in2 : P2in1;
        std::vector<AreaQuery> qs; Csr c;
        int rc = projected_candidates(h, dir == 0 ? KF2 : KF1, P, th, 0, 0, nullptr, &qs, &c); if (rc) return rc;
        std::vector<int>& out = dir == 0 ? vnMatch1 : vnMatch2;
        for (int i = 0; i < P->M; i++) {
            if (!qs[i].active) continue;
            int bestDist = 0x7fffffff, bestIdx = -1;
            for (int k = 0; k < c.count[i]; k++) {
                const int idx = c.ent[2 * (size_t)(c.start[i] + k)], dist = c.ent[2 * (size_t)(c.start[i] + k) + 1] & 0xFFFF;
                if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
            }
            if (bestDist <= TH_HIGH) out[i] = bestIdx;
        }
    }
    int nfound = 0;
    for (int i1 = 0; i1 < N1; i1++) {                                // :1918-1932
        matches12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { matches12[i1] = idx2; nfound++; }
    }
    if (nfound_out) *nfound_out = nfound;
    return ORBX_OK;
}

int orbm_distinctive_descriptors(orbx_extractor* h, const uint8_t* desc, const int* start, int P, int* best) {
    if (!h || P < 0 || (P > 0 && (!start || !best))) return fail(ORBX_E_ARG, "null");
    if (P == 0) return ORBX_OK;
    rt::set_device(h->device);
    const int total = start[P];
    if (total < 0 || (total > 0 && !desc)) return fail(ORBX_E_ARG, "bad descriptor list");
    for (int p = 0; p < P; p++) if (start[p + 1] < start[p]) return fail(ORBX_E_ARG, "start[] must be non-decreasing");
    int e = upload(h, SR_DESC, desc, 32 * (size_t)total) | h->d_si[SI_QSTART].ensure(P + 1) | h->d_si[SI_BEST].ensure(P);
    if (!e) e = rt::copy_h2d(h->d_si[SI_QSTART].p, start, sizeof(int) * (size_t)(P + 1), h->s0);
    if (e) return fail(ORBX_E_DEVICE, "upload/allocation failed");
    dim3 grid((P + 3) / 4, 1, 1), blk(256, 1, 1);
    ORBX_LAUNCH(k_distinctive, grid, blk, 0, h->s0, (const unsigned long long*)h->d_sr[SR_DESC].p, (const int*)h->d_si[SI_QSTART].p, P, h->d_si[SI_BEST].p);
    if (fetch_sync(h, best, h->d_si[SI_BEST].p, sizeof(int) * (size_t)P) || rt::check_launch()) return fail(ORBX_E_DEVICE, "distinctive-descriptor kernel failed: %s", rt::last_error());
    return ORBX_OK;
}

// ---- two-camera (Frame::Nleft != -1) branches.  Each camera's window search runs on the device (one k_area_search per camera); the
// reference interleaves the two cameras per map point and its assignments feed the occupancy test of later points, so the accept
// logic is replayed here over both candidate lists in the reference's order.
namespace {
int camera_candidates(orbx_extractor* h, const OrbmFrameView* F, const std::vector<AreaQuery>& qs, const uint8_t* qdesc, Csr* c) {
    DeviceFrame D;
    int rc = upload_frame(h, F, &D); if (rc) return rc;
    return run_area_search(h, D, qs, qdesc, c);
}
inline int cand_idx(const Csr& c, int q, int k) { return c.ent[2 * (size_t)(c.start[q] + k)]; }
inline int cand_dl(const Csr& c, int q, int k) { return c.ent[2 * (size_t)(c.start[q] + k) + 1]; }
}  // namespace

int orbm_search_by_projection_mappoints_fisheye(orbx_extractor* h, const OrbmFisheyeFrameView* F, const OrbmMapPointView* P,
                                                const OrbmMapPointRightView* PR, float th, int far_points, float th_far, float nnratio,
                                                int* assigned, int* nmatches_out) {
    if (!h || !F || !P || !PR || !assigned) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    const int M = P->M, NL = F->left.N, NR = F->right.N;
    const bool bFactor = th != 1.0;
    std::vector<AreaQuery> ql(M), qr(M);
    std::vector<uint8_t> alive(M > 0 ? M : 1, 0);
    for (int i = 0; i < M; i++) {
        memset(&ql[i], 0, sizeof(AreaQuery)); memset(&qr[i], 0, sizeof(AreaQuery));
        const bool inl = P->in_view[i] != 0, inr = PR->in_view_r[i] != 0;
        if (!inl && !inr) continue;                                       // :53-60
        if (far_points && P->track_depth[i] > th_far) continue;
        if (P->is_bad[i]) continue;
        alive[i] = 1;
        if (inl) {
            const int lvl = P->scale_level[i];
            if (lvl >= 0 && lvl < F->left.nlevels) {
                float r = P->view_cos[i] > 0.998 ? 2.5f : 4.0f;
                if (bFactor) r *= th;
                AreaQuery& q = ql[i];
                q.x = P->proj_x[i]; q.y = P->proj_y[i]; q.r = r * F->left.scale_factors[lvl]; q.min_level = lvl - 1; q.max_level = lvl; q.active = 1;
            }
        }
        if (inr) {
            const int lvl = PR->scale_level_r[i];
            if (lvl != -1 && lvl >= 0 && lvl < F->right.nlevels) {            // :172-173
                const float r = PR->view_cos_r[i] > 0.998 ? 2.5f : 4.0f;     // the right pass does not apply th (:174)
                AreaQuery& q = qr[i];
                q.x = PR->proj_xr[i]; q.y = PR->proj_yr[i]; q.r = r * F->right.scale_factors[lvl]; q.min_level = lvl - 1; q.max_level = lvl; q.active = 1;
            }
        }
    }
    Csr cl, cr;
    int rc = camera_candidates(h, &F->left, ql, P->desc, &cl); if (rc) return rc;
    rc = camera_candidates(h, &F->right, qr, P->desc, &cr); if (rc) return rc;
    std::vector<uint8_t> occ((size_t)NL + NR + 1, 0);
    if (F->left.occupied) memcpy(occ.data(), F->left.occupied, NL);
    if (F->right.occupied) memcpy(occ.data() + NL, F->right.occupied, NR);
    for (int i = 0; i < NL + NR; i++) assigned[i] = -1;
    int nmatches = 0;
    for (int i = 0; i < M; i++) {
        if (!alive[i]) continue;
        const uint8_t obs = P->has_obs ? P->has_obs[i] : 1;
        bool skip_right = false;
        if (ql[i].active && cl.count[i] > 0) {                               // :62-166
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            for (int k = 0; k < cl.count[i]; k++) {
                const int idx = cand_idx(cl, i, k), dl = cand_dl(cl, i, k);
                if (occ[idx]) continue;
                const int dist = dl & 0xFFFF, level = dl >> 16;
                if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = level; bestIdx = idx; }
                else if (dist < bestDist2) { bestLevel2 = level; bestDist2 = dist; }
            }
            if (bestDist <= TH_HIGH) {
                if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) skip_right = true;     // `continue` of the reference: next map point
                else if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                    assigned[bestIdx] = i; occ[bestIdx] = obs;
                    const int ltr = F->left_to_right ? F->left_to_right[bestIdx] : -1;
                    if (ltr != -1) { assigned[ltr + NL] = i; occ[ltr + NL] = obs; nmatches++; }
                    nmatches++;
                }
            }
        }
        if (skip_right) continue;
        if (qr[i].active) {                                                  // :170-236
            if (cr.count[i] == 0) continue;
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            for (int k = 0; k < cr.count[i]; k++) {
                const int idx = cand_idx(cr, i, k), dl = cand_dl(cr, i, k);
                if (occ[idx + NL]) continue;
                const int dist = dl & 0xFFFF, level = dl >> 16;
                if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = level; bestIdx = idx; }
                else if (dist < bestDist2) { bestLevel2 = level; bestDist2 = dist; }
            }
            if (bestDist <= TH_HIGH) {
                if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
                const int rtl = F->right_to_left ? F->right_to_left[bestIdx] : -1;
                if (rtl != -1) { assigned[rtl] = i; occ[rtl] = obs; nmatches++; }
                assigned[bestIdx + NL] = i; occ[bestIdx + NL] = obs;
                nmatches++;
            }
        }
    }
    if (nmatches_out) *nmatches_out = nmatches;
    return ORBX_OK;
}

int orbm_search_local_points_fisheye(orbx_extractor* h, const OrbmFisheyeFrameView* F, const OrbmFrustumRigView* V, const OrbmWorldPointView* P, float cos_limit, float th,
                                     int far_points, float th_far, float nnratio, const OrbmTrackOut* left, const OrbmTrackOutRight* right, int* assigned, int* nmatches_out) {
    if (!h || !F || !V || !P || !assigned) return fail(ORBX_E_ARG, "null");
    if (P->M < 0 || (P->M > 0 && (!P->pos || !P->normal || !P->min_distance || !P->max_distance || !P->desc))) return fail(ORBX_E_ARG, "bad map point view");
    const size_t M1 = P->M > 0 ? P->M : 1;
    // the tracking fields of both cameras come back once (they are the reference's MapPoint members; the searches read them as views)
    std::vector<uint8_t> inl(M1), inr(M1), zero(M1, 0), one(M1, 1);
    std::vector<float> f(9 * M1);
    std::vector<int> ll(M1), lr(M1);
    OrbmTrackOut L = {inl.data(), f.data(), f.data() + M1, f.data() + 2 * M1, f.data() + 3 * M1, f.data() + 4 * M1, ll.data()};
    OrbmTrackOutRight R = {inr.data(), f.data() + 5 * M1, f.data() + 6 * M1, f.data() + 7 * M1, f.data() + 8 * M1, lr.data()};
    int rc = orbm_is_in_frustum_rig(h, V, P, cos_limit, &L, &R); if (rc) return rc;
    OrbmMapPointView PV; memset(&PV, 0, sizeof PV);
    PV.M = P->M; PV.in_view = L.in_view; PV.proj_x = L.proj_x; PV.proj_y = L.proj_y; PV.proj_xr = L.proj_xr; PV.scale_level = L.scale_level; PV.view_cos = L.view_cos;
    PV.track_depth = L.depth; PV.is_bad = P->is_bad ? P->is_bad : zero.data(); PV.has_obs = P->has_obs ? P->has_obs : one.data(); PV.desc = P->desc;
    OrbmMapPointRightView PR; PR.in_view_r = R.in_view_r; PR.proj_xr = R.proj_xr; PR.proj_yr = R.proj_yr; PR.scale_level_r = R.scale_level_r; PR.view_cos_r = R.view_cos_r;
    rc = orbm_search_by_projection_mappoints_fisheye(h, F, &PV, &PR, th, far_points, th_far, nnratio, assigned, nmatches_out); if (rc) return rc;
    const size_t n = (size_t)(P->M > 0 ? P->M : 0);
    if (left) {
        if (left->in_view) memcpy(left->in_view, L.in_view, n); if (left->proj_x) memcpy(left->proj_x, L.proj_x, 4 * n); if (left->proj_y) memcpy(left->proj_y, L.proj_y, 4 * n);
        if (left->depth) memcpy(left->depth, L.depth, 4 * n); if (left->view_cos) memcpy(left->view_cos, L.view_cos, 4 * n); if (left->scale_level) memcpy(left->scale_level, L.scale_level, 4 * n);
        if (left->proj_xr) memcpy(left->proj_xr, L.proj_xr, 4 * n);
    }
    if (right) {
        if (right->in_view_r) memcpy(right->in_view_r, R.in_view_r, n); if (right->proj_xr) memcpy(right->proj_xr, R.proj_xr, 4 * n); if (right->proj_yr) memcpy(right->proj_yr, R.proj_yr, 4 * n);
        if (right->depth_r) memcpy(right->depth_r, R.depth_r, 4 * n); if (right->view_cos_r) memcpy(right->view_cos_r, R.view_cos_r, 4 * n);
        if (right->scale_level_r) memcpy(right->scale_level_r, R.scale_level_r, 4 * n);
    }
    return ORBX_OK;
}

int orbm_search_by_projection_frame_fisheye(orbx_extractor* h, const OrbmFisheyeFrameView* Cur, const OrbmLastFrameView* Last, const float* proj_ur,
                                            const float* proj_vr, float th, int forward, int backward, int check_ori, int* assigned, int* nmatches_out) {
    if (!h || !Cur || !Last || !proj_ur || !proj_vr || !assigned) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    const int NLast = Last->N, NL = Cur->left.N, NR = Cur->right.N;
    std::vector<AreaQuery> ql(NLast), qr(NLast);
    for (int i = 0; i < NLast; i++) {
        memset(&ql[i], 0, sizeof(AreaQuery)); memset(&qr[i], 0, sizeof(AreaQuery));
        if (!Last->valid[i]) continue;
        const float u = Last->proj_u[i], v = Last->proj_v[i];
        if (u < Cur->left.min_x || u > Cur->left.max_x) continue;          // :2003-2006
        if (v < Cur->left.min_y || v > Cur->left.max_y) continue;
        const int oct = Last->octave[i];
        if (oct < 0 || oct >= Cur->left.nlevels) continue;
        for (int cam = 0; cam < 2; cam++) {
            AreaQuery& q = cam == 0 ? ql[i] : qr[i];
            q.x = cam == 0 ? u : proj_ur[i]; q.y = cam == 0 ? v : proj_vr[i]; q.r = th * Cur->left.scale_factors[oct];
            if (forward) { q.min_level = oct; q.max_level = -1; }
            else if (backward) { q.min_level = 0; q.max_level = oct; }
            else { q.min_level = oct - 1; q.max_level = oct + 1; }
            q.active = 1;
        }
    }
    Csr cl, cr;
    int rc = camera_candidates(h, &Cur->left, ql, Last->desc, &cl); if (rc) return rc;
    rc = camera_candidates(h, &Cur->right, qr, Last->desc, &cr); if (rc) return rc;
    std::vector<uint8_t> occ((size_t)NL + NR + 1, 0);
    if (Cur->left.occupied) memcpy(occ.data(), Cur->left.occupied, NL);
    if (Cur->right.occupied) memcpy(occ.data() + NL, Cur->right.occupied, NR);
    for (int i = 0; i < NL + NR; i++) assigned[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    for (int i = 0; i < NLast; i++) {
        if (!ql[i].active || cl.count[i] == 0) continue;                   // an empty left window also skips the right camera (:2025-2026)
        const uint8_t obs = Last->has_obs ? Last->has_obs[i] : 1;
        {
            int bestDist = 256, bestIdx2 = -1;
            for (int k = 0; k < cl.count[i]; k++) {
                const int i2 = cand_idx(cl, i, k), dist = cand_dl(cl, i, k) & 0xFFFF;
                if (occ[i2]) continue;
                if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
            }
            if (bestDist <= TH_HIGH) {
                assigned[bestIdx2] = i; occ[bestIdx2] = obs; nmatches++;
                if (check_ori) rotHist[rot_bin(Last->angle[i], Cur->left.keys_un[bestIdx2].angle)].push_back(bestIdx2);
            }
        }
        {
            int bestDist = 256, bestIdx2 = -1;
            for (int k = 0; k < cr.count[i]; k++) {
                const int i2 = cand_idx(cr, i, k), dist = cand_dl(cr, i, k) & 0xFFFF;
                if (occ[i2 + NL]) continue;
                if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
            }
            if (bestDist <= TH_HIGH) {
                assigned[bestIdx2 + NL] = i; occ[bestIdx2 + NL] = obs; nmatches++;
                if (check_ori) rotHist[rot_bin(Last->angle[i], Cur->right.keys_un[bestIdx2].angle)].push_back(bestIdx2 + NL);
            }
        }
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { assigned[idx] = -2; nmatches--; }
    }
    if (nmatches_out) *nmatches_out = nmatches;
    return ORBX_OK;
}

}  // extern "C"
