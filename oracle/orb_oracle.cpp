// oracle/orb_oracle.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle).  Never linked into, imported by or
// executed from the product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it.
//
// Plain-C++ restatement of the reference ORB front-end hot path,
// written independently of the reference sources and *validated against them*: oracle/_ref/
// (the reference's own src/ORBextractor.cc compiled against oracle/opencv_shim) must produce
// identical keypoints/descriptors (tests/test_oracle_reference.py).
//
// PARITY STATUS: pinned against the reference's own extractor source for everything the reference
// implements itself (tables, cell grid, threshold fallback, quadtree, IC-angle, steered BRIEF,
// lapping reorder).  UNPINNED for the OpenCV primitives underneath (resize / FAST / GaussianBlur /
// fastAtan2), because OpenCV is not vendored in the reference, not installed here, and the
// reference ships no golden vectors.  Those live in orb_primitives.h.
//
// The extractor restatement deliberately uses the *derived* formulation the GPU kernels use
// (SURVEY.md §8a row F2): one FAST score map at min(iniTh,minTh), cell-local strict 3x3 NMS, and a
// per-cell threshold choice — so that "restatement == reference" proves the derivation.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "orb_primitives.h"
static const signed char kBriefPattern[1024] = {
#include "../orb_slam3_detailed_comments_amd/csrc/brief_pattern.inc"
};

namespace {

struct Kp { float x, y, size, angle, response; int octave, class_id; };   // cv::KeyPoint fields, 28 B
struct Cand { int x, y, score; };                                          // x,y relative to minBorder (16)

const int kPatch = 31, kHalfPatch = 15, kEdge = 19;   // src/ORBextractor.cc:76-78

struct Level {
    int w = 0, h = 0, quota = 0;
    float scale = 1.f, inv_scale = 1.f, sigma2 = 1.f, inv_sigma2 = 1.f;
    std::vector<uint8_t> img, blur;
    std::vector<Cand> cands;
    std::vector<Kp> kps;      // after quadtree + orientation, level coordinates
    bool blurred = false;
};

struct Oracle {
    int nfeatures, nlevels, iniTh, minTh, gauss_variant;
    double scaleFactor;
    std::vector<Level> lv;
    int umax[16];
};

// ---- E0: constructor tables (src/ORBextractor.cc:468-571) ---------------------------------------
void init_tables(Oracle& o) {
    o.lv.resize(o.nlevels);
    o.lv[0].scale = 1.f; o.lv[0].sigma2 = 1.f;
    for (int i = 1; i < o.nlevels; i++) {
        o.lv[i].scale = (float)(o.lv[i - 1].scale * o.scaleFactor);   // float * double member
        o.lv[i].sigma2 = o.lv[i].scale * o.lv[i].scale;
    }
    for (int i = 0; i < o.nlevels; i++) {
        o.lv[i].inv_scale = 1.0f / o.lv[i].scale;
        o.lv[i].inv_sigma2 = 1.0f / o.lv[i].sigma2;
    }
    float factor = (float)(1.0f / o.scaleFactor);
    float nDesired = o.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)o.nlevels));
    int sum = 0;
    for (int l = 0; l < o.nlevels - 1; l++) {
        o.lv[l].quota = orbp::round_half_even(nDesired);
        sum += o.lv[l].quota;
        nDesired *= factor;
    }
    o.lv[o.nlevels - 1].quota = std::max(o.nfeatures - sum, 0);
    // umax (:542-570)
    int vmax = orbp::floor_i(kHalfPatch * sqrt(2.f) / 2 + 1);
    int vmin = orbp::ceil_i(kHalfPatch * sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) o.umax[v] = orbp::round_half_even(sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (o.umax[v0] == o.umax[v0 + 1]) ++v0;
        o.umax[v] = v0;
        ++v0;
    }
}

// ---- E2: pyramid (src/ORBextractor.cc:1687-1738); borders are never consumed, so levels are dense --
void build_pyramid(Oracle& o, const uint8_t* img, int w, int h, int stride) {
    for (int l = 0; l < o.nlevels; l++) {
        Level& L = o.lv[l];
        L.w = orbp::round_half_even((float)w * L.inv_scale);
        L.h = orbp::round_half_even((float)h * L.inv_scale);
        L.img.assign((size_t)L.w * L.h, 0);
        L.blurred = false;
        if (l == 0) for (int y = 0; y < h; y++) memcpy(&L.img[(size_t)y * w], img + (size_t)y * stride, w);
        else orbp::resize_linear_u8(o.lv[l - 1].img.data(), o.lv[l - 1].w, o.lv[l - 1].h, o.lv[l - 1].w,
                                    L.img.data(), L.w, L.h, L.w);
    }
}

// ---- E3 (as F2): per-cell FAST candidates (src/ORBextractor.cc:1061-1166) ---------------------------
bool fast_candidates(const Oracle& o, Level& L) {
    L.cands.clear();
    const float W = 35;
    const int minBX = kEdge - 3, minBY = minBX, maxBX = L.w - kEdge + 3, maxBY = L.h - kEdge + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return false;
    const int wCell = (int)ceil(width / nCols), hCell = (int)ceil(height / nRows);
    // score map at the lower threshold (0 where not a corner)
    const int t0 = std::min(o.iniTh, o.minTh);
    std::vector<uint8_t> S((size_t)L.w * L.h, 0);
    for (int y = minBY + 3; y < maxBY - 3; y++)
        for (int x = minBX + 3; x < maxBX - 3; x++) {
            const uint8_t* p = &L.img[(size_t)y * L.w + x];
            if (orbp::fast_is_corner(p, L.w, t0)) S[(size_t)y * L.w + x] = (uint8_t)orbp::fast_corner_score(p, L.w, t0);
        }
    std::vector<Cand> hi, lo;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxX > maxBX) maxX = (float)maxBX;
            // detectable interior of the cell window; neighbours outside it count as score 0
            const int x0 = (int)iniX + 3, x1 = (int)maxX - 3, y0 = (int)iniY + 3, y1 = (int)maxY - 3;
            hi.clear(); lo.clear();
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) {
                    const int s = S[(size_t)y * L.w + x];
                    if (s < std::min(o.iniTh, o.minTh) || s == 0) continue;
                    bool ismax = true;
                    for (int dy = -1; dy <= 1 && ismax; dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            if (!dx && !dy) continue;
                            const int xx = x + dx, yy = y + dy;
                            const int q = (xx >= x0 && xx < x1 && yy >= y0 && yy < y1) ? S[(size_t)yy * L.w + xx] : 0;
                            if (!(s > q)) { ismax = false; break; }
                        }
                    if (!ismax) continue;
                    Cand c = {x - minBX, y - minBY, s};
                    if (s >= o.iniTh) hi.push_back(c);
                    if (s >= o.minTh) lo.push_back(c);
                }
            const std::vector<Cand>& use = hi.empty() ? lo : hi;
            L.cands.insert(L.cands.end(), use.begin(), use.end());
        }
    }
    return true;
}

// ---- E4: quadtree distribution (src/ORBextractor.cc:602-697, :711-1057) -----------------------------
struct QNode { int x0, y0, x1, y1; std::vector<int> keys; bool noMore = false; };

void divide(const QNode& n, const std::vector<Cand>& c, QNode ch[4]) {
    const int halfX = (int)ceil(static_cast<float>(n.x1 - n.x0) / 2);
    const int halfY = (int)ceil(static_cast<float>(n.y1 - n.y0) / 2);
    const int mx = n.x0 + halfX, my = n.y0 + halfY;
    ch[0] = {n.x0, n.y0, mx, my, {}, false};
    ch[1] = {mx, n.y0, n.x1, my, {}, false};
    ch[2] = {n.x0, my, mx, n.y1, {}, false};
    ch[3] = {mx, my, n.x1, n.y1, {}, false};
    for (int k : n.keys) {
        const bool left = (float)c[k].x < (float)mx, top = (float)c[k].y < (float)my;
        ch[left ? (top ? 0 : 2) : (top ? 1 : 3)].keys.push_back(k);
    }
    for (int q = 0; q < 4; q++) if (ch[q].keys.size() == 1) ch[q].noMore = true;
}

// Returns selected candidate indices in the reference's list order.
bool quadtree(const std::vector<Cand>& c, int width, int height, int N, std::vector<int>& result) {
    result.clear();
    const int nIni = (int)round(static_cast<float>(width) / height);
    if (nIni < 1) return false;
    const float hX = static_cast<float>(width) / nIni;
    std::vector<QNode> nodes;          // arena
    std::vector<int> order;            // the std::list, front first
    for (int i = 0; i < nIni; i++) {
        QNode n; n.x0 = (int)(hX * static_cast<float>(i)); n.x1 = (int)(hX * static_cast<float>(i + 1)); n.y0 = 0; n.y1 = height;
        nodes.push_back(n);
    }
    for (size_t k = 0; k < c.size(); k++) nodes[(int)((float)c[k].x / hX)].keys.push_back((int)k);
    for (int i = 0; i < nIni; i++) {
        if (nodes[i].keys.size() == 1) { nodes[i].noMore = true; order.push_back(i); }
        else if (!nodes[i].keys.empty()) order.push_back(i);
    }
    bool finish = false;
    std::vector<std::pair<int, int>> toExpand;   // (nkeys, node id), creation order
    auto cmp = [&](const std::pair<int, int>& a, const std::pair<int, int>& b) {
        if (a.first < b.first) return true;
        if (a.first > b.first) return false;
        return nodes[a.second].x0 < nodes[b.second].x0;
    };
    while (!finish) {
        const int prevSize = (int)order.size();
        int nToExpand = 0;
        toExpand.clear();
        std::vector<int> front;                      // newest first
        std::vector<int> kept;
        for (int id : order) {
            if (nodes[id].noMore) { kept.push_back(id); continue; }
            QNode ch[4]; divide(nodes[id], c, ch);
            std::vector<int> block;
            for (int q = 0; q < 4; q++) if (!ch[q].keys.empty()) {
                nodes.push_back(ch[q]);
                const int cid = (int)nodes.size() - 1;
                block.push_back(cid);
                if (nodes[cid].keys.size() > 1) { nToExpand++; toExpand.push_back({(int)nodes[cid].keys.size(), cid}); }
            }
            std::reverse(block.begin(), block.end());             // push_front n1..n4 => n4 first
            front.insert(front.begin(), block.begin(), block.end());
        }
        order = front; order.insert(order.end(), kept.begin(), kept.end());
        if ((int)order.size() >= N || (int)order.size() == prevSize) finish = true;
        else if ((int)order.size() + nToExpand * 3 > N) {
            while (!finish) {
                const int prev2 = (int)order.size();
                std::vector<std::pair<int, int>> prev = toExpand;
                toExpand.clear();
                std::sort(prev.begin(), prev.end(), cmp);
                std::vector<char> erased(nodes.size() + 4 * prev.size() + 8, 0);
                std::vector<int> front2;
                int size = prev2;
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    QNode ch[4]; divide(nodes[prev[j].second], c, ch);
                    std::vector<int> block;
                    for (int q = 0; q < 4; q++) if (!ch[q].keys.empty()) {
                        nodes.push_back(ch[q]);
                        const int cid = (int)nodes.size() - 1;
                        block.push_back(cid);
                        if (nodes[cid].keys.size() > 1) toExpand.push_back({(int)nodes[cid].keys.size(), cid});
                    }
                    std::reverse(block.begin(), block.end());
                    front2.insert(front2.begin(), block.begin(), block.end());
                    erased[prev[j].second] = 1;
                    size += (int)block.size() - 1;
                    if (size >= N) break;
                }
                std::vector<int> next = front2;
                for (int id : order) if (!erased[id]) next.push_back(id);
                order = next;
                if ((int)order.size() >= N || (int)order.size() == prev2) finish = true;
            }
        }
    }
    for (int id : order) {
        const std::vector<int>& keys = nodes[id].keys;
        int best = keys[0];
        for (size_t k = 1; k < keys.size(); k++) if (c[keys[k]].score > c[best].score) best = keys[k];
        result.push_back(best);
    }
    return true;
}

// ---- E5: IC_Angle (src/ORBextractor.cc:91-138) ------------------------------------------------------
float ic_angle(const Oracle& o, const Level& L, float px, float py) {
    int m01 = 0, m10 = 0;
    const uint8_t* center = &L.img[(size_t)orbp::round_half_even(py) * L.w + orbp::round_half_even(px)];
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * center[u];
    const int step = L.w;
    for (int v = 1; v <= kHalfPatch; ++v) {
        int v_sum = 0;
        const int d = o.umax[v];
        for (int u = -d; u <= d; ++u) {
            const int vp = center[u + v * step], vm = center[u - v * step];
            v_sum += (vp - vm);
            m10 += u * (vp + vm);
        }
        m01 += v * v_sum;
    }
    return orbp::fast_atan2_deg((float)m01, (float)m10);
}

// ---- E7: steered BRIEF (src/ORBextractor.cc:150-203) ------------------------------------------------
void brief(const Level& L, const Kp& kp, uint8_t* desc) {
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float angle = kp.angle * factorPI;
    const float a = cosf(angle), b = sinf(angle);
    const uint8_t* center = &L.blur[(size_t)orbp::round_half_even(kp.y) * L.w + orbp::round_half_even(kp.x)];
    const int step = L.w;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            const signed char* p = &kBriefPattern[4 * (8 * i + j)];
            const int t0 = center[orbp::round_half_even(p[0] * b + p[1] * a) * step + orbp::round_half_even(p[0] * a - p[1] * b)];
            const int t1 = center[orbp::round_half_even(p[2] * b + p[3] * a) * step + orbp::round_half_even(p[2] * a - p[3] * b)];
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

// ---- M0: DescriptorDistance (src/ORBmatcher.cc:2383-2403) — SWAR popcount over 8 x int32 -------------
int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y; memcpy(&x, a + 4 * i, 4); memcpy(&y, b + 4 * i, 4);
        uint32_t v = x ^ y;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

}  // namespace

extern "C" {

void* orbo_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh, int gauss_variant) {
    Oracle* o = new Oracle;
    o->nfeatures = nfeatures; o->scaleFactor = scaleFactor; o->nlevels = nlevels; o->iniTh = iniTh; o->minTh = minTh;
    o->gauss_variant = gauss_variant;
    init_tables(*o);
    return o;
}
void orbo_destroy(void* h) { delete (Oracle*)h; }

void orbo_tables(void* h, int* quotas, int* umax16, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2) {
    Oracle* o = (Oracle*)h;
    for (int i = 0; i < o->nlevels; i++) {
        quotas[i] = o->lv[i].quota; scale[i] = o->lv[i].scale; inv_scale[i] = o->lv[i].inv_scale;
        sigma2[i] = o->lv[i].sigma2; inv_sigma2[i] = o->lv[i].inv_sigma2;
    }
    for (int i = 0; i < 16; i++) umax16[i] = o->umax[i];
}

// ORBextractor::operator() (src/ORBextractor.cc:1557-1682).  Returns monoIndex, -1 on empty image,
// -2 if the image is too small for the 35-px cell grid (the reference would divide by zero).
int orbo_extract(void* h, const uint8_t* img, int w, int hgt, int stride, int lap0, int lap1,
                 void* kps_out, uint8_t* desc_out, int cap, int* n_out) {
    Oracle* o = (Oracle*)h;
    *n_out = 0;
    if (!img || w <= 0 || hgt <= 0) return -1;
    build_pyramid(*o, img, w, hgt, stride);
    int total = 0;
    for (int l = 0; l < o->nlevels; l++) {
        Level& L = o->lv[l];
        L.kps.clear();
        if (!fast_candidates(*o, L)) return -2;
        std::vector<int> sel;
        if (!quadtree(L.cands, (L.w - kEdge + 3) - (kEdge - 3), (L.h - kEdge + 3) - (kEdge - 3), L.quota, sel)) return -2;
        const int scaledPatch = (int)(kPatch * L.scale);
        for (int k : sel) {
            Kp kp; kp.x = (float)(L.cands[k].x + kEdge - 3); kp.y = (float)(L.cands[k].y + kEdge - 3);
            kp.size = (float)scaledPatch; kp.angle = -1; kp.response = (float)L.cands[k].score; kp.octave = l; kp.class_id = -1;
            L.kps.push_back(kp);
        }
        total += (int)L.kps.size();
    }
    for (int l = 0; l < o->nlevels; l++)
        for (Kp& kp : o->lv[l].kps) kp.angle = ic_angle(*o, o->lv[l], kp.x, kp.y);
    *n_out = total;
    if (total > cap) return -3;
    Kp* out = (Kp*)kps_out;
    int mono = 0, stereo = total - 1;
    for (int l = 0; l < o->nlevels; l++) {
        Level& L = o->lv[l];
        if (L.kps.empty()) continue;
        L.blur.resize(L.img.size());
        orbp::gaussian_blur7_u8(L.img.data(), L.w, L.h, L.w, L.blur.data(), L.w, o->gauss_variant);
        L.blurred = true;
        for (const Kp& k0 : L.kps) {
            uint8_t d[32]; brief(L, k0, d);
            Kp kp = k0;
            if (l != 0) { kp.x = kp.x * L.scale; kp.y = kp.y * L.scale; }
            int idx;
            if (kp.x >= lap0 && kp.x <= lap1) idx = stereo--; else idx = mono++;
            out[idx] = kp;
            memcpy(desc_out + 32 * (size_t)idx, d, 32);
        }
    }
    return mono;
}

int orbo_level_info(void* h, int level, int* w, int* hgt, int* quota, float* scale) {
    Oracle* o = (Oracle*)h; const Level& L = o->lv[level];
    *w = L.w; *hgt = L.h; *quota = L.quota; *scale = L.scale; return 0;
}
int orbo_level_image(void* h, int level, uint8_t* dst) { Oracle* o = (Oracle*)h; const Level& L = o->lv[level]; memcpy(dst, L.img.data(), L.img.size()); return 0; }
int orbo_level_blurred(void* h, int level, uint8_t* dst) {
    Oracle* o = (Oracle*)h; Level& L = o->lv[level];
    if (!L.blurred) { L.blur.resize(L.img.size()); orbp::gaussian_blur7_u8(L.img.data(), L.w, L.h, L.w, L.blur.data(), L.w, o->gauss_variant); L.blurred = true; }
    memcpy(dst, L.blur.data(), L.blur.size()); return 0;
}
int orbo_level_candidates(void* h, int level, int* xys, int cap) {
    Oracle* o = (Oracle*)h; const Level& L = o->lv[level];
    for (size_t i = 0; i < L.cands.size() && (int)i < cap; i++) { xys[3 * i] = L.cands[i].x; xys[3 * i + 1] = L.cands[i].y; xys[3 * i + 2] = L.cands[i].score; }
    return (int)L.cands.size();
}
int orbo_level_keypoints(void* h, int level, void* out, int cap) {
    Oracle* o = (Oracle*)h; const Level& L = o->lv[level];
    for (size_t i = 0; i < L.kps.size() && (int)i < cap; i++) ((Kp*)out)[i] = L.kps[i];
    return (int)L.kps.size();
}
// Stand-alone quadtree on a caller-supplied candidate list (x,y,score triples), for GPU stage tests.
int orbo_quadtree(const int* xys, int n, int width, int height, int N, int* sel_out, int cap) {
    std::vector<Cand> c(n);
    for (int i = 0; i < n; i++) c[i] = {xys[3 * i], xys[3 * i + 1], xys[3 * i + 2]};
    std::vector<int> sel;
    if (!quadtree(c, width, height, N, sel)) return -2;
    for (size_t i = 0; i < sel.size() && (int)i < cap; i++) sel_out[i] = sel[i];
    return (int)sel.size();
}

int orbo_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

// ---- M1: Frame::ComputeStereoMatches (src/Frame.cc:1102-1358) ---------------------------------------
// hL/hR: oracle handles whose last orbo_extract() built the left/right pyramids.
int orbo_stereo_matches(void* hL, void* hR, const void* kpsL_, int nL, const uint8_t* descL,
                        const void* kpsR_, int nR, const uint8_t* descR, float mbf, float mb,
                        float* uRight, float* depth) {
    Oracle* oL = (Oracle*)hL; Oracle* oR = (Oracle*)hR;
    const Kp* kL = (const Kp*)kpsL_; const Kp* kR = (const Kp*)kpsR_;
    const int TH_HIGH = 100, TH_LOW = 50;
    for (int i = 0; i < nL; i++) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = oL->lv[0].h;
    std::vector<std::vector<int>> rows(nRows);
    for (int iR = 0; iR < nR; iR++) {
        const float kpY = kR[iR].y;
        const float r = 2.0f * oL->lv[kR[iR].octave].scale;
        const int maxr = (int)ceil(kpY + r), minr = (int)floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) rows[yi].push_back(iR);
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < nL; iL++) {
        const Kp& kpL = kL[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        if ((int)vL < 0 || (int)vL >= nRows) continue;
        const std::vector<int>& cand = rows[(int)vL];
        if (cand.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH; int bestIdxR = 0;
        for (int iR : cand) {
            const Kp& kpR = kR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = descriptor_distance(descL + 32 * (size_t)iL, descR + 32 * (size_t)iR);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = kR[bestIdxR].x;
            const float scaleFactor = oL->lv[kpL.octave].inv_scale;
            const float scaleduL = roundf(kpL.x * scaleFactor), scaledvL = roundf(kpL.y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const Level& PL = oL->lv[kpL.octave]; const Level& PR = oR->lv[kpL.octave];
            const uint8_t* IL = &PL.img[(size_t)((int)scaledvL - w) * PL.w + ((int)scaleduL - w)];
            int bestD = INT_MAX, bestinc = 0;
            float vD[2 * 5 + 1];
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= PR.w) continue;
            for (int inc = -L; inc <= L; inc++) {
                const uint8_t* IR = &PR.img[(size_t)((int)scaledvL - w) * PR.w + ((int)scaleduR0 + inc - w)];
                float dist = (float)orbp::norm_l1_u8(IL, PL.w, IR, PR.w, 2 * w + 1, 2 * w + 1);
                if (dist < bestD) { bestD = (int)dist; bestinc = inc; }
                vD[L + inc] = dist;
            }
            if (bestinc == -L || bestinc == L) continue;
            const float d1 = vD[L + bestinc - 1], d2 = vD[L + bestinc], d3 = vD[L + bestinc + 1];
            const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = oL->lv[kpL.octave].scale * ((float)scaleduR0 + (float)bestinc + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                depth[iL] = mbf / disparity;
                uRight[iL] = bestuR;
                vDistIdx.push_back({bestD, iL});
            }
        }
    }
    if (vDistIdx.empty()) return 0;   // the reference indexes an empty vector here (UB); guarded
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    int kept = (int)vDistIdx.size();
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        uRight[vDistIdx[i].second] = -1; depth[vDistIdx[i].second] = -1; kept--;
    }
    return kept;
}

// ---- M2 (matching part): BFMatcher(NORM_HAMMING).knnMatch(k=2) + Lowe ratio (src/Frame.cc:1553-1562) ---
// For every query row: two smallest distances, strict '<' insertion so equal distances keep the lower
// train index first.  ratio_ok[q] = d0 < 0.7*d1 evaluated in double like the reference.
void orbo_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx0, int* d0, int* idx1, int* d1, uint8_t* ratio_ok) {
    for (int i = 0; i < nq; i++) {
        int b0 = INT_MAX, b1 = INT_MAX, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            const int d = descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
            else if (d < b1) { b1 = d; i1 = j; }
        }
        idx0[i] = i0; idx1[i] = i1; d0[i] = i0 < 0 ? -1 : b0; d1[i] = i1 < 0 ? -1 : b1;
        ratio_ok[i] = (i1 >= 0 && (float)b0 < (float)b1 * 0.7) ? 1 : 0;
    }
}

// ---- raw primitive entry points (tests/test_oracle_primitives.py cross-checks them against slow definitions) ----
int orbo_prim_fast(const uint8_t* img, int w, int h, int step, int threshold, int nonmax, int* xys, int cap) {
    std::vector<orbp::FastPoint> pts;
    orbp::fast9_16(img, w, h, (size_t)step, threshold, nonmax != 0, pts);
    for (size_t i = 0; i < pts.size() && (int)i < cap; i++) { xys[3 * i] = pts[i].x; xys[3 * i + 1] = pts[i].y; xys[3 * i + 2] = pts[i].score; }
    return (int)pts.size();
}
int orbo_prim_is_corner(const uint8_t* p, int step, int threshold) { return orbp::fast_is_corner(p, (size_t)step, threshold) ? 1 : 0; }
int orbo_prim_corner_score(const uint8_t* p, int step, int threshold) { return orbp::fast_corner_score(p, (size_t)step, threshold); }
void orbo_prim_resize(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) { orbp::resize_linear_u8(src, sw, sh, (size_t)sw, dst, dw, dh, (size_t)dw); }
void orbo_prim_blur(const uint8_t* src, int w, int h, uint8_t* dst, int variant) { orbp::gaussian_blur7_u8(src, w, h, (size_t)w, dst, (size_t)w, variant); }
void orbo_prim_border(const uint8_t* src, int w, int h, uint8_t* dst, int b) { orbp::make_border_reflect101(src, w, h, (size_t)w, dst, (size_t)(w + 2 * b), b, b, b, b); }
int orbo_prim_round(double v) { return orbp::round_half_even(v); }

// glibc cosf/sinf, exposed so tests can pin the device-side model (csrc/glibc_sincosf_model.h).
float orbo_cosf(float x) { return cosf(x); }
float orbo_sinf(float x) { return sinf(x); }
float orbo_fast_atan2(float y, float x) { return orbp::fast_atan2_deg(y, x); }

}  // extern "C"
