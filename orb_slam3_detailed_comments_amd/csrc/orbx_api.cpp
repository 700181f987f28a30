// orbx_api.cpp — host side of liborbx_hip.so: parameter tables (reference ORBextractor constructor,
// src/ORBextractor.cc:468-571), per-resolution geometry (pyramid sizes :1691-1692, FAST cell grid
// :1069-1129, resize coefficients of cv::resize), device memory, stream orchestration and the C ABI of
// include/orbx.h.  Compiled by hipcc for gfx950 (product) or by g++ against tests/emu (tests only).
#include <mutex>
#include "orbx_internal.h"

#ifndef ORBX_PYR_FUSED_BATCH
#define ORBX_PYR_FUSED_BATCH 8     // k_pyramid_fused (all levels in one launch) for batches up to this many images
#endif
#ifndef ORBX_RESIZE_MIN_BLOCKS
#define ORBX_RESIZE_MIN_BLOCKS 2048 // k_resize_rows: strips are halved until a level gives at least this many workgroups (tests lower it to keep long strips at small batches)
#endif
#ifndef ORBX_RESIZE_STRIP
#define ORBX_RESIZE_STRIP 16        // k_resize_rows: output rows per wave for large batches
#endif
#ifndef ORBX_FAST_LIST_BYTES
#define ORBX_FAST_LIST_BYTES 2048   // k_fast_cells: 1024 list entries (corners of the cell so far + survivors waiting for their score)
#endif
#ifndef ORBX_QT_BIG_PIXELS
#define ORBX_QT_BIG_PIXELS 40000     // levels whose detection area has at least this many pixels get 1024 quadtree threads at small batches (tests also build 0).
                                     // 150000 left level 2 of a 752x480 pyramid (6 k candidates) on 256 threads: the longest tree of a stereo pair, 66 us against level 0's 57
#endif
#ifndef ORBX_QT_WIDE_BATCH
#define ORBX_QT_WIDE_BATCH 32        // batches of up to this many images run the quadtree of their large levels on 1024 threads
#endif
#ifndef ORBX_PRESORT_MAX
#define ORBX_PRESORT_MAX 5      // deepest quadtree level resolved by the up-front counting sort (tests also build 0 and 2)
#endif

using namespace orbx;

namespace orbx {
static thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf; return code;
}
const char* last_error_string() { return g_err.c_str(); }
}  // namespace orbx

namespace {
const char* kStageNames[ORBX_NSTAGES] = {"import", "pyramid", "fast_cells", "quadtree", "blur", "layout", "orient_brief", "match"};
}

namespace {

// ---- E0: tables of the reference constructor -------------------------------------------------------
void init_tables(orbx_extractor* h) {
    h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
    for (int i = 1; i < h->nlevels; i++) {
        h->scale[i] = (float)(h->scale[i - 1] * h->scaleFactor);
        h->sigma2[i] = h->scale[i] * h->scale[i];
    }
    for (int i = 0; i < h->nlevels; i++) {
        h->inv_scale[i] = 1.0f / h->scale[i];
        h->inv_sigma2[i] = 1.0f / h->sigma2[i];
    }
    const float factor = (float)(1.0f / h->scaleFactor);
    float want = h->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)h->nlevels));
    int sum = 0;
    for (int l = 0; l < h->nlevels - 1; l++) {
        h->quota[l] = round_half_even_f(want);
        sum += h->quota[l];
        want *= factor;
    }
    h->quota[h->nlevels - 1] = std::max(h->nfeatures - sum, 0);
    // circular patch row half-widths
    int* um = h->umax.u;
    const int vmax = (int)floor(kHalfPatch * sqrt(2.f) / 2 + 1), vmin = (int)ceil(kHalfPatch * sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) um[v] = round_half_even_d(sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (um[v0] == um[v0 + 1]) ++v0;
        um[v] = v0;
        ++v0;
    }
}

// cv::resize(INTER_LINEAR, 8U) coefficient tables for one axis
void resize_axis(int ssize, int dsize, bool clamp_edges, std::vector<ResizeTap>& out) {
    const double inv = (double)dsize / ssize, sc = 1.0 / inv;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * sc - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (clamp_edges) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        int a0 = round_half_even_f((1.f - f) * 2048.f), a1 = round_half_even_f(f * 2048.f);
        a0 = std::min(std::max(a0, -32768), 32767); a1 = std::min(std::max(a1, -32768), 32767);
        ResizeTap t; t.ofs = s; t.w = (a0 & 0xFFFF) | (a1 << 16);
        out.push_back(t);
    }
}

// k_pyramid_fused: per tile column (row) of the top level and per level the region a workgroup computes and the interval it owns.
// Ownership goes down the levels along the left (upper) taps: a tile owns at level l what starts at the left tap of the first pixel it owns
// at level l + 1, so the intervals of neighbouring tiles meet and cover the level; the region is what the region of the level above reads,
// widened to the owned interval (scale factors > 2 leave pixels that the next level never reads) and, for columns, to whole dwords.
void pyramid_spans(const orbx_extractor* h, bool columns, std::vector<PyrSpan>& out, int& ntiles, int* max_extent) {
    const int nl = h->nlevels, top = nl - 1;
    auto size_of = [&](int l) { return columns ? h->lv[l].w : h->lv[l].h; };
    auto taps_of = [&](int l) { return (columns ? h->xtab.data() + h->lv[l].xtab_off : h->ytab.data() + h->lv[l].ytab_off); };   // level l from level l - 1
    auto lo = [&](int l, int d) { return std::min(std::max(taps_of(l)[d].ofs, 0), size_of(l - 1) - 1); };
    auto hi = [&](int l, int d) { return std::min(std::max(taps_of(l)[d].ofs + 1, 0), size_of(l - 1) - 1); };
    const int T = (size_of(top) + kPyrTile - 1) / kPyrTile;
    ntiles = T;
    for (int l = 0; l < nl; l++) max_extent[l] = 0;
    out.assign((size_t)T * nl, PyrSpan{0, 0, 0, 0});
    std::vector<int> own0((size_t)(T + 1) * nl);
    for (int t = 0; t <= T; t++) own0[(size_t)t * nl + top] = std::min(t * kPyrTile, size_of(top));
    for (int l = top - 1; l >= 0; l--)
        for (int t = 0; t <= T; t++)
            own0[(size_t)t * nl + l] = t == 0 ? 0 : t == T ? size_of(l) : lo(l + 1, own0[(size_t)t * nl + l + 1]);
    for (int t = 0; t < T; t++) {
        int a = 0, b = 0;
        for (int l = top; l >= 0; l--) {
            const int o0 = own0[(size_t)t * nl + l], o1 = own0[(size_t)(t + 1) * nl + l];
            if (l == top) { a = o0; b = o1; }
            else {
                const int na = lo(l + 1, a), nb = hi(l + 1, std::min(b, size_of(l + 1)) - 1) + 1;
                a = l == 0 ? na : std::min(o0, na); b = l == 0 ? nb : std::max(o1, nb);
            }
            if (columns) { a &= ~3; b = (b + 3) & ~3; }
            PyrSpan sp; sp.a = (int16_t)a; sp.b = (int16_t)b; sp.o0 = (int16_t)o0; sp.o1 = (int16_t)o1;
            out[(size_t)t * nl + l] = sp;
            max_extent[l] = std::max(max_extent[l], b - a);
        }
    }
}

// How the quadtree stage is launched (k_quadtree.hip).  A tree's node arrays take 81 bytes per node; beside them the bucket offsets, the per-segment
// bucket counters (counter_bytes per bucket) and the two coordinate tables of the level.  Levels first_lds .. nlevels-1 run in the LDS form with
// node_cap_lds nodes each; levels 0 .. first_lds-1 (the big quotas of a 5 x nFeatures extractor) keep their node arrays in the global pool.
struct QuadtreePlan { int first_lds, node_cap_lds, lut_x, lut_y; size_t smem_lds, smem_spill, pool_stride; };
QuadtreePlan quadtree_plan(const orbx_extractor* h, int counter_bytes) {
    QuadtreePlan p;
    p.lut_x = (int)align_up((size_t)h->lv[0].bw + 1, 8); p.lut_y = (int)align_up((size_t)h->lv[0].bh + 1, 8);       // level 0 is the largest
    const size_t tables = (size_t)(h->nb_cap + 2) * 4 + (size_t)counter_bytes * h->nb_cap + 2 * (size_t)(p.lut_x + p.lut_y) + 64;
    const size_t limit = rt::lds_limit(h->device);
    p.first_lds = h->nlevels; p.node_cap_lds = 0; p.smem_lds = 0;
    int cap = 0;
    for (int l = h->nlevels - 1; l >= 0; l--) {
        cap = std::max(cap, h->lv[l].kp_cap + 8);
        const size_t smem = align_up((size_t)cap * 81, 16) + tables;
        if (cap > h->qt_lds_nodes || smem + 2048 > limit) break;
        p.first_lds = l; p.node_cap_lds = cap; p.smem_lds = smem;
    }
    p.smem_spill = tables;
    p.pool_stride = align_up((size_t)h->node_cap * 81, 256);
    return p;
}

int configure(orbx_extractor* h, int W, int H, int B) {
    if (W <= 0 || H <= 0 || B <= 0) return fail(ORBX_E_ARG, "bad size %dx%d batch %d", W, H, B);
    const bool same_geom = (W == h->W && H == h->H);
    if (same_geom && B <= h->maxB && h->cfg_qt_lds_nodes == h->qt_lds_nodes) return ORBX_OK;
    if (rt::set_device(h->device)) return fail(ORBX_E_DEVICE, "hipSetDevice(%d) failed", h->device);
    if (!same_geom) {
        if (W - 2 * kBorder > 4095 || H - 2 * kBorder > 4095) return fail(ORBX_E_ARG, "image larger than 4127 px is not supported");
        // the tables below are rebuilt in place; until they are complete (and uploaded) the handle has no geometry, so a rejected size
        // cannot leave host tables of one resolution beside device tables of another
        h->W = h->H = 0; h->maxB = 0; h->lastB = 0; h->ncells = 0; h->kp_total_cap = 0;
        h->cells.clear(); h->xtab.clear(); h->ytab.clear();
        size_t off = 0; int cand_off = 0, kp_off = 0, node_cap = 0, tile_b = 0, inner_b = 0, nb_cap = 1;
        for (int l = 0; l < h->nlevels; l++) {
            LevelInfo& L = h->lv[l];
            memset(&L, 0, sizeof L);
            L.w = round_half_even_f((float)W * h->inv_scale[l]);     // src/ORBextractor.cc:1692
            L.h = round_half_even_f((float)H * h->inv_scale[l]);
            L.pitch = (int)align_up((size_t)L.w, 64);
            L.off = (int)off; off += align_up((size_t)L.pitch * L.h, 256);
            L.scale = h->scale[l]; L.inv_scale = h->inv_scale[l]; L.quota = h->quota[l];
            L.patch = (int)(31 * h->scale[l]);                       // :1184
            // FAST cell grid, :1069-1129
            const float Wc = 35;
            const int minBX = kBorder, minBY = kBorder, maxBX = L.w - kBorder, maxBY = L.h - kBorder;
            const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
            if (width < Wc || height < Wc) return fail(ORBX_E_ARG, "level %d (%dx%d) is too small for the 35-px FAST cell grid", l, L.w, L.h);
            L.ncols = (int)(width / Wc); L.nrows = (int)(height / Wc);
            L.wcell = (int)ceil(width / L.ncols); L.hcell = (int)ceil(height / L.nrows);
            L.bw = maxBX - minBX; L.bh = maxBY - minBY;
            L.nini = (int)roundf((float)L.bw / (float)L.bh);         // :718
            if (L.nini < 1) return fail(ORBX_E_ARG, "aspect ratio < 0.5 is not supported (the reference divides by zero)");
            L.hX = (float)L.bw / (float)L.nini;                      // :722
            L.cell_begin = (int)h->cells.size();
            L.cand_off = cand_off;
            int slot = cand_off;
            for (int i = 0; i < L.nrows; i++) {
                const float iniY = (float)(minBY + i * L.hcell);
                float maxY = iniY + L.hcell + 6;
                if (iniY >= maxBY - 3) continue;
                if (maxY > maxBY) maxY = (float)maxBY;
                for (int j = 0; j < L.ncols; j++) {
                    const float iniX = (float)(minBX + j * L.wcell);
                    float maxX = iniX + L.wcell + 6;
                    if (iniX >= maxBX - 6) continue;
                    if (maxX > maxBX) maxX = (float)maxBX;
                    CellInfo c; memset(&c, 0, sizeof c);
                    c.level = (int16_t)l;
                    c.x0 = (int16_t)((int)iniX + 3); c.x1 = (int16_t)((int)maxX - 3);
                    c.y0 = (int16_t)((int)iniY + 3); c.y1 = (int16_t)((int)maxY - 3);
                    c.slot_off = slot;
                    const int iw = std::max(c.x1 - c.x0, 0), ih = std::max(c.y1 - c.y0, 0);
                    slot += ((iw + 1) / 2) * ((ih + 1) / 2);         // max number of strict 3x3 local maxima
                    if (iw > 0 && ih > 0) {
                        const int gx0 = (c.x0 - 3) & ~3, gx1 = (c.x1 + 3 + 3) & ~3;                 // dword-aligned window (k_fast_cells)
                        const int wp = gx1 - gx0 <= kFastPitch ? kFastPitch : gx1 - gx0;             // LDS pitch of the window tile
                        if (iw > 240 || iw * ih >= 8192 || wp * (ih + 6) > 16384) return fail(ORBX_E_ARG, "FAST cell too large");   // k_fast_cells: one row per trip, 14-bit tile offsets
                        tile_b = std::max(tile_b, wp * (ih + 6));
                        inner_b = std::max(inner_b, wp * (ih + 2));                                  // score tile
                    }
                    h->cells.push_back(c);
                }
            }
            L.cell_count = (int)h->cells.size() - L.cell_begin;
            L.cand_cap = slot - cand_off;
            cand_off = slot;
            // a full pass never overshoots the quota, a final-round split adds at most 3 (:912, :1006)
            L.kp_cap = std::max(L.quota + 3, 4 * L.nini);
            L.kp_off = kp_off; kp_off += L.kp_cap;
            // presort depth of the quadtree kernel: nini * 4^D buckets, at most 1024
            L.presort_depth = 0;
            while (L.presort_depth < ORBX_PRESORT_MAX && L.nini * (1 << (2 * (L.presort_depth + 1))) <= 1024) L.presort_depth++;
            nb_cap = std::max(nb_cap, L.nini * (1 << (2 * L.presort_depth)));
            L.qt_threads = (long long)L.bw * L.bh >= ORBX_QT_BIG_PIXELS ? kQuadtreeThreads : 256;
            node_cap = std::max(node_cap, L.kp_cap + 8);
            if (l > 0) {
                L.xtab_off = (int)h->xtab.size(); resize_axis(h->lv[l - 1].w, L.w, true, h->xtab);
                L.ytab_off = (int)h->ytab.size(); resize_axis(h->lv[l - 1].h, L.h, false, h->ytab);
                // k_resize_rows: both taps of output columns x .. x+3 (x a multiple of 4) within 8 bytes of the first one's left tap
                bool ok = true;
                const ResizeTap* xt = h->xtab.data() + L.xtab_off;
                for (int x = 0; x < L.w && ok; x += 4)
                    for (int k = 0; k < 4; k++) { const int xx = std::min(x + k, L.w - 1); if (std::min(xt[xx].ofs + 1, h->lv[l - 1].w - 1) - xt[x].ofs > 7 || xt[xx].ofs < xt[x].ofs) ok = false; }
                h->resize_rows_ok[l] = ok;
            }
        }
        h->pyr_stride = off; h->cand_stride = (size_t)cand_off; h->ncells = (int)h->cells.size();
        h->pyr_fused_ok = false; h->xspan.clear(); h->yspan.clear();
        if (h->nlevels > 1) {
            int mw[kMaxLevels], mh[kMaxLevels];
            pyramid_spans(h, true, h->xspan, h->pyr_ntx, mw);
            pyramid_spans(h, false, h->yspan, h->pyr_nty, mh);
            int nt = 0;
            for (int l = 1; l < h->nlevels; l++) { h->pyr_toff.x[l] = nt; nt += mw[l]; h->pyr_toff.y[l] = nt; nt += mh[l]; }
            h->pyr_toff.total = nt;
            // LDS: regions of the even levels in one buffer, of the odd levels in the other; a region must stay below 2^13 dwords (the kernel's
            // division by multiplication) and the whole below the 64 KB every device grants
            size_t bufa = 0, bufb = 0, items = 0;
            for (int tx = 0; tx < h->pyr_ntx; tx++)
                for (int ty = 0; ty < h->pyr_nty; ty++)
                    for (int l = 0; l < h->nlevels; l++) {
                        const PyrSpan& sx = h->xspan[(size_t)tx * h->nlevels + l]; const PyrSpan& sy = h->yspan[(size_t)ty * h->nlevels + l];
                        const size_t bytes = (size_t)(sx.b - sx.a) * (sy.b - sy.a);
                        (l & 1 ? bufb : bufa) = std::max(l & 1 ? bufb : bufa, bytes);
                        items = std::max(items, bytes / 4);
                    }
            h->pyr_buf_a = (int)align_up(bufa, 16); h->pyr_buf_b = (int)align_up(bufb, 16);
            h->pyr_fused_ok = items < 8192 && (size_t)h->pyr_buf_a + h->pyr_buf_b + 8 * (size_t)nt + 48 * (size_t)h->nlevels <= 60000;
        }
        h->kp_total_cap = kp_off; h->node_cap = node_cap; h->nb_cap = nb_cap;
        h->fast_tile_bytes = (int)align_up((size_t)tile_b, 16); h->fast_inner_bytes = (int)align_up((size_t)inner_b, 16);
        if (h->kp_total_cap >= 65535) { h->kp_total_cap = 0; return fail(ORBX_E_ARG, "nfeatures too large"); }
        int e = 0;
        e |= h->d_lv.ensure(kMaxLevels); e |= h->d_cells.ensure(h->cells.size());
        e |= h->d_xtab.ensure(std::max<size_t>(h->xtab.size(), 1)); e |= h->d_ytab.ensure(std::max<size_t>(h->ytab.size(), 1));
        e |= h->d_xspan.ensure(std::max<size_t>(h->xspan.size(), 1)); e |= h->d_yspan.ensure(std::max<size_t>(h->yspan.size(), 1));
        if (e) return fail(ORBX_E_DEVICE, "device allocation failed (tables)");
        rt::copy_h2d(h->d_lv.p, h->lv, sizeof(LevelInfo) * kMaxLevels, h->s0);
        rt::copy_h2d(h->d_cells.p, h->cells.data(), sizeof(CellInfo) * h->cells.size(), h->s0);
        if (!h->xtab.empty()) rt::copy_h2d(h->d_xtab.p, h->xtab.data(), sizeof(ResizeTap) * h->xtab.size(), h->s0);
        if (!h->ytab.empty()) rt::copy_h2d(h->d_ytab.p, h->ytab.data(), sizeof(ResizeTap) * h->ytab.size(), h->s0);
        if (!h->xspan.empty()) rt::copy_h2d(h->d_xspan.p, h->xspan.data(), sizeof(PyrSpan) * h->xspan.size(), h->s0);
        if (!h->yspan.empty()) rt::copy_h2d(h->d_yspan.p, h->yspan.data(), sizeof(PyrSpan) * h->yspan.size(), h->s0);
        if (rt::stream_sync(h->s0)) return fail(ORBX_E_DEVICE, "table upload failed: %s", rt::last_error());
        h->W = W; h->H = H;                                        // commit
    }
    if (B > h->maxB || h->cfg_qt_lds_nodes != h->qt_lds_nodes) {
        // node pool of the quadtree levels beyond the LDS: sized for the wider counters (small batches), which leave the LDS form the fewest levels
        const QuadtreePlan qp = quadtree_plan(h, 32);
        if (qp.first_lds > 0 && h->d_qtpool.ensure((size_t)std::max(B, h->maxB) * qp.first_lds * qp.pool_stride))
            return fail(ORBX_E_DEVICE, "device allocation failed (quadtree node pool, batch %d of %dx%d)", B, W, H);
        h->cfg_qt_lds_nodes = h->qt_lds_nodes;
    }
    if (B > h->maxB) {
        const size_t b = (size_t)B, cap = (size_t)h->kp_total_cap;
        int e = 0;
        e |= h->d_pyr.ensure(b * h->pyr_stride + 256); e |= h->d_blur.ensure(b * h->pyr_stride + 256);
        e |= h->d_slots.ensure(b * h->cand_stride + 4); e |= h->d_candA.ensure(b * h->cand_stride + 4); e |= h->d_candB.ensure(b * h->cand_stride + 4);
        e |= h->d_cell_count.ensure(b * h->ncells); e |= h->d_lvl_count.ensure(b * h->nlevels);
        e |= h->d_lvl_keys.ensure(b * cap); e |= h->d_final_idx.ensure(b * cap);
        // counts [b] | mono indices [b] | status word (4 ints): one block, so that orbx_fetch brings all three back in one copy; d_status is a
        // view of its tail (never allocated or freed by itself)
        e |= h->d_nm.ensure(2 * b + 4);
        h->d_status.p = h->d_nm.p ? h->d_nm.p + 2 * b : nullptr; h->d_status.n = h->d_nm.p ? 4 : 0;
        e |= h->d_kps.ensure(b * cap); e |= h->d_desc.ensure(b * cap * 4); e |= h->d_aux.ensure(b * cap * 4);
        e |= h->d_uRight.ensure(b * cap); e |= h->d_depth.ensure(b * cap); e |= h->d_sad.ensure(b * cap); e |= h->d_nmatch.ensure(b);
        e |= h->d_rowstart.ensure(b * (size_t)((h->H >> kStereoRowShift) + 3)); e |= h->d_rowitems.ensure(b * cap);
        e |= h->d_knn.ensure(4 * b * cap); e |= h->d_ratio.ensure(b * cap);
        e |= h->h_nm.ensure(3 * b + 4); e |= h->d_qtprof.ensure(32);
        if (e) return fail(ORBX_E_DEVICE, "device allocation failed (batch %d of %dx%d)", B, W, H);
        rt::memset_async(h->d_status.p, 0, 4 * sizeof(int), h->s0);
        rt::memset_async(h->d_pyr.p, 0, b * h->pyr_stride + 256, h->s0);     // defined row padding for frames written in place (orbx_input_buffer)
        h->maxB = B;
    }
    return ORBX_OK;
}

void stage_begin(orbx_extractor* h, int st, rt::stream_t s) { if (h->profile) rt::event_record(h->ev_stage[st][0], s); }
void stage_end(orbx_extractor* h, int st, rt::stream_t s) { if (h->profile) rt::event_record(h->ev_stage[st][1], s); }

// level 0 from raw camera frames: geometry (remap / resize, per channel) first, grey conversion second - the order of the reference
// (System::TrackStereo feeds the rectified / resized frame to Tracking::GrabImage*, which converts it)
void enqueue_input(orbx_extractor* h, int B, const uint8_t* d_images, int sw, int sh, int stride, size_t image_stride) {
    const LevelInfo& L0 = h->lv[0];
    const dim3 blk2(64, 4, 1);
    const int C = h->in_channels;
    uint8_t* lvl0 = h->d_pyr.p + L0.off;
    const uint8_t* cur = d_images; int cur_stride = stride; size_t cur_img = image_stride;
    if (h->in_geometry != 0) {
        const bool direct = C == 1;                               // single channel: straight into the pyramid
        uint8_t* dst = direct ? lvl0 : h->d_frame.p;
        const int dpitch = direct ? L0.pitch : L0.w * C; const size_t dstride = direct ? h->pyr_stride : (size_t)L0.w * L0.h * C;
        dim3 grid(((dpitch + C - 1) / C + 63) / 64, (L0.h + 3) / 4, B);
        if (h->in_geometry == 1)
            ORBX_LAUNCH(k_input_remap, grid, blk2, 0, h->s0, cur, sw, sh, cur_stride, cur_img, C, (const float*)h->d_mapx.p, (const float*)h->d_mapy.p,
                        L0.w, L0.h, dst, dpitch, dstride);
        else
            ORBX_LAUNCH(k_input_resize, grid, blk2, 0, h->s0, cur, sw, sh, cur_stride, cur_img, C, (const ResizeTap*)h->d_in_xt.p,
                        (const ResizeTap*)h->d_in_yt.p, L0.w, L0.h, dst, dpitch, dstride);
        cur = dst; cur_stride = dpitch; cur_img = dstride;
        if (direct) return;
    }
    // cv::cvtColor 8U: OpenCV 4.x (RY15, GY15, BY15, 15) or 3.x (R2Y, G2Y, B2Y, yuv_shift = 14)
    const int ry = h->in_gray_variant ? 4899 : 9798, gy = h->in_gray_variant ? 9617 : 19235, by = h->in_gray_variant ? 1868 : 3735;
    dim3 grid((L0.pitch + 255) / 256, (L0.h + 3) / 4, B);      // four pixels per thread
    ORBX_LAUNCH(k_input_gray, grid, blk2, 0, h->s0, cur, cur_stride, cur_img, C, h->in_rgb ? 0 : 2, ry, gy, by, h->in_gray_variant ? 14 : 15, L0.w, L0.h,
                lvl0, L0.pitch, h->pyr_stride);
}

int enqueue_extract(orbx_extractor* h, int B, const uint8_t* d_images, int src_w, int src_h, int stride, size_t image_stride, int lap0, int lap1) {
    const int nl = h->nlevels;
    const dim3 blk2(64, 4, 1), blk1(256, 1, 1);
    // (no fill launches in front of the chain: the quadtree's capacity flag is cleared by k_fast_cells, and the descriptor rows beyond n[b] -
    // which read as zero: fixed-shape blocks for collectives - by k_orient_brief, one row per unused keypoint slot)
    stage_begin(h, ST_IMPORT, h->s0);
    if (h->in_active && (h->in_channels != 1 || h->in_geometry != 0)) enqueue_input(h, B, d_images, src_w, src_h, stride, image_stride);
    else if (d_images == h->d_pyr.p + h->lv[0].off && stride == h->lv[0].pitch && image_stride == h->pyr_stride) {
        // the caller wrote the frames into level 0 itself (orbx_input_buffer): nothing to import
    } else {
        const LevelInfo& L0 = h->lv[0];
        dim3 grid((L0.pitch + 255) / 256, (L0.h + 3) / 4, B);
        ORBX_LAUNCH(k_import, grid, blk2, 0, h->s0, (const LevelInfo*)h->d_lv.p, d_images, stride, image_stride, h->d_pyr.p, h->pyr_stride);
    }
    stage_end(h, ST_IMPORT, h->s0);
    // the input images have been consumed: an upload into the caller's buffer may start (orbx_device_upload_async).  Large batches record it here
    // (the upload of batch i + 1 overlaps the rest of batch i); small ones on demand (orbx_internal.h: a record costs the next kernel ~6 us)
    if (B > ORBX_QT_WIDE_BATCH) { rt::event_record(h->ev_import, h->s0); h->import_lazy = false; }
    else h->import_lazy = true;
    stage_begin(h, ST_PYRAMID, h->s0);
    // small batches: every level in one launch (a chain of dependent launches costs 4.5 us per level whatever the work); large batches:
    // the streaming kernel per level (a third of the instructions per pixel, and its launches hide behind the other handles' kernels)
    const bool fused = nl > 1 && h->pyr_fused_ok && (h->pyramid_mode == 2 || (h->pyramid_mode == 0 && B <= ORBX_PYR_FUSED_BATCH));
    if (fused) {
        dim3 grid(h->pyr_ntx * h->pyr_nty, B, 1);
        const size_t smem = (size_t)h->pyr_buf_a + h->pyr_buf_b + 8 * (size_t)h->pyr_toff.total + 48 * (size_t)nl;
        ORBX_LAUNCH(k_pyramid_fused, grid, dim3(kPyrThreads, 1, 1), smem, h->s0, (const LevelInfo*)h->d_lv.p, nl, (const ResizeTap*)h->d_xtab.p, (const ResizeTap*)h->d_ytab.p,
                    (const PyrSpan*)h->d_xspan.p, (const PyrSpan*)h->d_yspan.p, h->pyr_ntx, h->d_pyr.p, h->pyr_stride, h->pyr_buf_a, h->pyr_buf_b, h->pyr_toff);
    }
    for (int l = 1; l < nl && !fused; l++) {
        const LevelInfo& L = h->lv[l]; const LevelInfo& S = h->lv[l - 1];
        if (h->resize_rows_ok[l]) {          // streaming form: the taps of 4 adjacent output columns fit an 8-byte source window
            // rows per wave: long strips share more source rows (a strip of n rows computes ~1.2 n + 1 horizontal rows), short ones give
            // small batches enough waves to hide the row loads
            int strip = ORBX_RESIZE_STRIP;
            while (strip > 4 && (long)((L.pitch + 255) / 256) * ((L.h + 4 * strip - 1) / (4 * strip)) * B < ORBX_RESIZE_MIN_BLOCKS) strip >>= 1;
            dim3 gridr((L.pitch + 255) / 256, (L.h + 4 * strip - 1) / (4 * strip), B);
            ORBX_LAUNCH(k_resize_rows, gridr, blk2, 0, h->s0, (const LevelInfo*)h->d_lv.p, l, (const ResizeTap*)h->d_xtab.p, (const ResizeTap*)h->d_ytab.p,
                        h->d_pyr.p, h->pyr_stride, strip);
            continue;
        }
        dim3 grid((L.pitch + 255) / 256, (L.h + kResizeRows - 1) / kResizeRows, B);
        // LDS window: source span of a 256 x 8 output tile (+ alignment and the +1 neighbour)
        const int lds_pitch = (int)align_up((size_t)ceil(256.0 * S.w / L.w) + 12, 4);
        const int lds_rows = (int)ceil((double)kResizeRows * S.h / L.h) + 3;
        ORBX_LAUNCH(k_resize, grid, blk2, (size_t)lds_pitch * lds_rows, h->s0, (const LevelInfo*)h->d_lv.p, l, (const ResizeTap*)h->d_xtab.p,
                    (const ResizeTap*)h->d_ytab.p, h->d_pyr.p, h->pyr_stride, lds_pitch, lds_rows);
    }
    stage_end(h, ST_PYRAMID, h->s0);
    BlurTaps taps;
    {
        static const int A[7] = {18, 34, 48, 56, 48, 34, 18}, Bt[7] = {18, 34, 49, 55, 49, 34, 18};
        for (int i = 0; i < 7; i++) taps.k[i] = h->gauss_variant == 1 ? Bt[i] : A[i];
    }
    // strips of 16 rows for small batches (more waves; the fused blur + FAST launch), of 32 rows for large ones (fewer halo rows)
    const bool blur_large = B > ORBX_QT_WIDE_BATCH;
    const int blur_rows = blur_large ? kBlurRowsLarge : kBlurRows;
    BlurTiles tiles; int nt = 0;
    for (int l = 0; l < nl; l++) { tiles.begin[l] = nt; nt += ((h->lv[l].w + 255) / 256) * ((h->lv[l].h + 4 * blur_rows - 1) / (4 * blur_rows)); }
    for (int l = nl; l <= kMaxLevels; l++) tiles.begin[l] = nt;
    const int fast_blocks = (h->ncells + 8 * kFastXcdRun - 1) / (8 * kFastXcdRun) * (8 * kFastXcdRun);   // whole XCD runs (k_fast_cells)
    // pad | window tile | score tile | u16 list: corners found so far + pending survivors of the quick test (scored whenever it fills up)
    const int list_bytes = ORBX_FAST_LIST_BYTES;
    const size_t fast_smem = 16 + (size_t)h->fast_tile_bytes + (size_t)h->fast_inner_bytes + (size_t)list_bytes + 64;
    const dim3 blkf(kFastThreadsDecl, 1, 1);
    // small batches: blur and FAST in one launch on one stream (k_fast_cells_blur: no fork / join).  The stage timers of the profiling modes keep
    // the two apart, so those run the large-batch form
    const bool small_forms = B <= ORBX_QT_WIDE_BATCH && !h->profile && !h->serial;
    const bool one_launch = small_forms && h->small_forms;
    if (one_launch) {
        dim3 grid(4 * nt + fast_blocks, B, 1);
        ORBX_LAUNCH(k_fast_cells_blur, grid, blkf, fast_smem, h->s0, (const LevelInfo*)h->d_lv.p, (const CellInfo*)h->d_cells.p, h->ncells,
                    (const uint8_t*)h->d_pyr.p, h->pyr_stride, h->iniTh, h->minTh, h->d_slots.p, h->cand_stride, h->d_cell_count.p,
                    h->fast_tile_bytes, list_bytes, h->d_status.p, nl, h->d_blur.p, taps, tiles, 4 * nt);
    } else {
        // fork: the blur only depends on the pyramid and runs beside FAST + quadtree on the second stream
        rt::stream_t sb = h->serial ? h->s0 : h->s1;    // serial profiling mode keeps every kernel on one stream
        rt::event_record(h->ev_fork, h->s0);
        rt::stream_wait_event(sb, h->ev_fork);
        stage_begin(h, ST_BLUR, sb);
        {
            dim3 grid(nt, B, 1);
            if (blur_large) ORBX_LAUNCH(k_blur_large, grid, blk2, 0, sb, (const LevelInfo*)h->d_lv.p, nl, (const uint8_t*)h->d_pyr.p, h->d_blur.p, h->pyr_stride, taps, tiles);
            else ORBX_LAUNCH(k_blur, grid, blk2, 0, sb, (const LevelInfo*)h->d_lv.p, nl, (const uint8_t*)h->d_pyr.p, h->d_blur.p, h->pyr_stride, taps, tiles);
        }
        stage_end(h, ST_BLUR, sb);
        rt::event_record(h->ev_join, sb);
        stage_begin(h, ST_FAST, h->s0);
        {
            dim3 grid(fast_blocks, B, 1);
            ORBX_LAUNCH(k_fast_cells, grid, blkf, fast_smem, h->s0, (const LevelInfo*)h->d_lv.p, (const CellInfo*)h->d_cells.p, h->ncells,
                        (const uint8_t*)h->d_pyr.p, h->pyr_stride, h->iniTh, h->minTh, h->d_slots.p, h->cand_stride, h->d_cell_count.p,
                        h->fast_tile_bytes, list_bytes, h->d_status.p);
        }
    }
    stage_end(h, ST_FAST, h->s0);
    // (+ the row index of every image's keypoints for ComputeStereoMatches: k_stereo_match reads the right image's)
    const int nb_rows = (h->H >> kStereoRowShift) + 2;
    stage_begin(h, ST_QUADTREE, h->s0);
    {
        // small batches: the large levels run on 1024 threads (half the latency of one tree); large batches: four waves per tree, more trees per CU
        int qt_block = 256;
        if (B <= ORBX_QT_WIDE_BATCH) for (int l = 0; l < nl; l++) qt_block = std::max(qt_block, h->lv[l].qt_threads);
        const int wide = qt_block > 256, counter_bytes = wide ? 32 : 16;
        const QuadtreePlan qp = quadtree_plan(h, counter_bytes);
        h->qt_pool_levels = qp.first_lds;
        const dim3 blkq(qt_block, 1, 1);
        long long* prof = h->serial ? (long long*)h->d_qtprof.p : (long long*)nullptr;
        if (qp.smem_spill + 2048 > rt::lds_limit(h->device))
            return fail(ORBX_E_CAPACITY, "the quadtree's bucket tables at %dx%d need %zu bytes of LDS, the device allows %zu", h->W, h->H, qp.smem_spill + 2048, rt::lds_limit(h->device));
        if (qp.first_lds > 0) {
            // the trees of the largest quotas first (they are the longest): node arrays in the pool, tables in LDS
            if (h->d_qtpool.n < (size_t)B * qp.first_lds * qp.pool_stride) return fail(ORBX_E_INTERNAL, "quadtree node pool not allocated");
            ORBX_LAUNCH(k_quadtree_spill, dim3(B, qp.first_lds, 1), blkq, qp.smem_spill, h->s0, (const LevelInfo*)h->d_lv.p, (const CellInfo*)h->d_cells.p, h->ncells,
                        (const int*)h->d_cell_count.p, (const uint32_t*)h->d_slots.p, h->cand_stride, h->d_candA.p, h->d_candB.p, h->cand_stride,
                        h->d_lvl_keys.p, h->kp_total_cap, h->d_lvl_count.p, nl, h->node_cap, h->nb_cap, qp.lut_x, qp.lut_y, h->d_status.p,
                        prof, wide, counter_bytes, h->d_qtpool.p, qp.pool_stride);
        }
        if (qp.first_lds < nl)
            ORBX_LAUNCH(k_quadtree, dim3(B, nl - qp.first_lds, 1), blkq, qp.smem_lds, h->s0, (const LevelInfo*)h->d_lv.p, (const CellInfo*)h->d_cells.p, h->ncells,
                        (const int*)h->d_cell_count.p, (const uint32_t*)h->d_slots.p, h->cand_stride, h->d_candA.p, h->d_candB.p, h->cand_stride,
                        h->d_lvl_keys.p, h->kp_total_cap, h->d_lvl_count.p, nl, qp.node_cap_lds, h->nb_cap, qp.lut_x, qp.lut_y, h->d_status.p,
                        prof, wide, counter_bytes, qp.first_lds);
    }
    stage_end(h, ST_QUADTREE, h->s0);
    stage_begin(h, ST_LAYOUT, h->s0);
    {
        dim3 grid(B, 1, 1);
        ORBX_LAUNCH(k_layout, grid, blk1, 2 * (size_t)(nb_rows + 1) * sizeof(int), h->s0, (const LevelInfo*)h->d_lv.p, nl, (const uint32_t*)h->d_lvl_keys.p, h->kp_total_cap,
                    (const int*)h->d_lvl_count.p, lap0, lap1, h->d_final_idx.p, h->d_nm.p, h->d_nm.p + h->maxB, nb_rows, h->d_rowstart.p, h->d_rowitems.p);
    }
    stage_end(h, ST_LAYOUT, h->s0);
    if (!one_launch) rt::stream_wait_event(h->s0, h->ev_join);
    stage_begin(h, ST_DESCRIBE, h->s0);
    {
        const bool small = B <= ORBX_QT_WIDE_BATCH;                 // latency-tuned variant: 2 keypoints per wave instead of 8
        const int kpw = small ? kKpPerWaveSmallDecl : kKpPerWaveDecl;
        const int gpi = (h->kp_total_cap + 4 * kpw - 1) / (4 * kpw);
        dim3 grid(gpi * 8 * ((B + 7) / 8), 1, 1);                  // an image's workgroups on one XCD (k_orient_brief)
        if (small) ORBX_LAUNCH(k_orient_brief_small, grid, blk1, 0, h->s0, (const LevelInfo*)h->d_lv.p, nl, (const uint8_t*)h->d_pyr.p, (const uint8_t*)h->d_blur.p,
                    h->pyr_stride, (const uint32_t*)h->d_lvl_keys.p, h->kp_total_cap, (const int*)h->d_lvl_count.p, (const int*)h->d_final_idx.p,
                    h->umax, h->d_kps.p, h->d_desc.p, (int4*)h->d_aux.p, B, gpi);
        else ORBX_LAUNCH(k_orient_brief, grid, blk1, 0, h->s0, (const LevelInfo*)h->d_lv.p, nl, (const uint8_t*)h->d_pyr.p, (const uint8_t*)h->d_blur.p,
                    h->pyr_stride, (const uint32_t*)h->d_lvl_keys.p, h->kp_total_cap, (const int*)h->d_lvl_count.p, (const int*)h->d_final_idx.p,
                    h->umax, h->d_kps.p, h->d_desc.p, (int4*)h->d_aux.p, B, gpi);
    }
    if (h->undist.active) {                                         // mvKeysUn (Frame::UndistortKeyPoints)
        dim3 grid((h->kp_total_cap + 255) / 256, B, 1);
        ORBX_LAUNCH(k_undistort, grid, blk1, 0, h->s0, (const KeyPointRec*)h->d_kps.p, (const int*)h->d_nm.p, h->kp_total_cap, h->undist, h->d_kps_un.p);
    }
    stage_end(h, ST_DESCRIBE, h->s0);
    h->done_lazy = true;                                            // ev_done: recorded by whoever waits for it (record_done_if_pending)
    h->lastB = B; h->extract_gen++; h->ex_undist_gen = h->undist_gen; h->ex_undist_active = h->undist.active != 0;
    if (rt::check_launch()) return fail(ORBX_E_DEVICE, "kernel launch failed: %s", rt::last_error());
    return ORBX_OK;
}

}  // namespace

extern "C" {

const char* orbx_last_error(void) { return orbx::last_error_string(); }
const char* orbx_stage_name(int i) { return (i >= 0 && i < ORBX_NSTAGES) ? kStageNames[i] : ""; }
int orbx_device_count(void) { return rt::device_count(); }

int orbx_create(orbx_extractor** out, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int device_id) {
    if (!out) return fail(ORBX_E_ARG, "null out");
    *out = nullptr;
    if (nfeatures <= 0 || nlevels < 1 || nlevels > kMaxLevels || !(scale_factor > 1.0f) || ini_th < 0 || min_th < 0 || ini_th > 255 || min_th > 255)
        return fail(ORBX_E_ARG, "bad extractor parameters");
    if (rt::device_count() <= device_id || device_id < 0) return fail(ORBX_E_DEVICE, "no usable GPU %d (HIP reports %d devices)", device_id, rt::device_count());
    if (rt::set_device(device_id)) return fail(ORBX_E_DEVICE, "hipSetDevice(%d) failed", device_id);
    orbx_extractor* h = new orbx_extractor();
    h->nfeatures = nfeatures; h->scaleFactor = scale_factor; h->nlevels = nlevels; h->iniTh = ini_th; h->minTh = min_th; h->device = device_id;
    init_tables(h);
    int e = rt::stream_create(&h->s0) | rt::stream_create(&h->s1) | rt::stream_create(&h->s_copy) | rt::event_create(&h->ev_fork) | rt::event_create(&h->ev_join) |
            rt::event_create(&h->ev_done) | rt::event_create(&h->ev_copy) | rt::event_create(&h->ev_import) | rt::event_create(&h->ev_lp);
    for (int i = 0; i < ORBX_NSTAGES; i++) { e |= rt::event_create(&h->ev_stage[i][0]); e |= rt::event_create(&h->ev_stage[i][1]); h->stage_ms[i] = 0; }
    h->have_streams = true;
    if (e) { orbx_destroy(h); return fail(ORBX_E_DEVICE, "stream/event creation failed"); }      // (gives back whatever was created)
    *out = h;
    return ORBX_OK;
}

void orbx_destroy(orbx_extractor* h) {
    if (!h) return;
    rt::set_device(h->device);
    if (h->have_streams) {
        rt::stream_sync(h->s0); rt::stream_sync(h->s1);
#ifndef ORBX_EMU
        if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
        if (h->graph) (void)hipGraphDestroy(h->graph);
#endif
        for (int i = 0; i < ORBX_NSTAGES; i++) { rt::event_destroy(h->ev_stage[i][0]); rt::event_destroy(h->ev_stage[i][1]); }
        rt::stream_sync(h->s_copy);
        rt::event_destroy(h->ev_fork); rt::event_destroy(h->ev_join); rt::event_destroy(h->ev_done); rt::event_destroy(h->ev_copy); rt::event_destroy(h->ev_import); rt::event_destroy(h->ev_lp);
        rt::stream_destroy(h->s0); rt::stream_destroy(h->s1); rt::stream_destroy(h->s_copy);
    }
    h->d_lv.release(); h->d_cells.release(); h->d_xtab.release(); h->d_ytab.release(); h->d_xspan.release(); h->d_yspan.release(); h->d_pyr.release(); h->d_blur.release(); h->d_stage.release();
    h->d_slots.release(); h->d_candA.release(); h->d_candB.release(); h->d_lvl_keys.release(); h->d_cell_count.release(); h->d_lvl_count.release();
    h->d_final_idx.release(); h->d_status.p = nullptr; h->d_status.n = 0; h->d_nm.release(); h->d_kps.release(); h->d_desc.release();
    h->d_uRight.release(); h->d_depth.release(); h->d_sad.release(); h->d_nmatch.release(); h->d_knn.release(); h->d_ratio.release();
    h->d_l2r.release(); h->d_r2l.release(); h->d_p3d.release(); h->d_hamA.release(); h->d_hamB.release(); h->d_hamOut.release(); h->h_stage.release(); h->h_nm.release();
    for (auto& x : h->d_sr) x.release();
    h->h_packA.release(); h->h_packB.release(); h->h_out.release(); h->h_res.release();
    for (auto& x : h->d_si) x.release();
    h->d_kps_un.release();
    h->d_lp.release(); h->d_depth_in.release(); h->h_lp_in.release(); h->h_lp_out.release();
    h->d_aux.release(); h->d_qtprof.release(); h->d_qtpool.release(); h->d_rowstart.release(); h->d_rowitems.release();
    h->d_mapx.release(); h->d_mapy.release(); h->d_in_xt.release(); h->d_in_yt.release(); h->d_frame.release();
    delete h;
}

int orbx_set_gaussian_taps(orbx_extractor* h, int variant) {
    if (!h || (variant != 0 && variant != 1)) return fail(ORBX_E_ARG, "bad gaussian variant");
    h->gauss_variant = variant; return ORBX_OK;
}
int orbx_reserve(orbx_extractor* h, int width, int height, int max_batch) { if (!h) return fail(ORBX_E_ARG, "null handle"); return configure(h, width, height, max_batch); }
int orbx_get_levels(const orbx_extractor* h) { return h ? h->nlevels : 0; }
float orbx_get_scale_factor(const orbx_extractor* h) { return h ? (float)h->scaleFactor : 0.f; }
int orbx_get_level_tables(const orbx_extractor* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* fpl, int* umax16) {
    if (!h) return fail(ORBX_E_ARG, "null handle");
    for (int i = 0; i < h->nlevels; i++) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->inv_scale[i];
        if (sigma2) sigma2[i] = h->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->inv_sigma2[i];
        if (fpl) fpl[i] = h->quota[i];
    }
    if (umax16) for (int i = 0; i < 16; i++) umax16[i] = h->umax.u[i];
    return ORBX_OK;
}
int orbx_max_keypoints(const orbx_extractor* h) {
    if (!h) return 0;
    if (h->kp_total_cap) return h->kp_total_cap;
    int s = 0; for (int l = 0; l < h->nlevels; l++) s += std::max(h->quota[l] + 3, 16); return s;   // before the geometry is known
}

int orbx_extract_batch(orbx_extractor* h, int B, const uint8_t* images, int width, int height, int stride, size_t image_stride,
                       int on_device, int lap0, int lap1) {
    if (!h) return fail(ORBX_E_ARG, "null handle");
    if (!images || width <= 0 || height <= 0 || B <= 0) return fail(ORBX_E_EMPTY, "empty image");
    const int C = h->in_active ? h->in_channels : 1;
    const size_t row_bytes = (size_t)width * C;
    if ((size_t)stride < row_bytes) return fail(ORBX_E_ARG, "stride < width * channels");
    if (B > 1 && image_stride < (size_t)stride * (height - 1) + row_bytes) return fail(ORBX_E_ARG, "image_stride too small");
    const bool geom = h->in_active && h->in_geometry != 0;
    int rc = configure(h, geom ? h->in_out_w : width, geom ? h->in_out_h : height, B);
    if (rc) return rc;
    rt::set_device(h->device);
    if (h->undist.active && h->d_kps_un.ensure((size_t)h->maxB * h->kp_total_cap)) return fail(ORBX_E_DEVICE, "allocation failed (undistorted keypoints)");
    if (h->in_active) {
        if (h->in_geometry == 2 && (h->in_tap_w != width || h->in_tap_h != height)) {        // cv::resize taps for this source size
            std::vector<ResizeTap> xt, yt;
            resize_axis(width, h->in_out_w, true, xt); resize_axis(height, h->in_out_h, false, yt);
            if (h->d_in_xt.ensure(xt.size()) || h->d_in_yt.ensure(yt.size())) return fail(ORBX_E_DEVICE, "allocation failed");
            rt::copy_h2d(h->d_in_xt.p, xt.data(), sizeof(ResizeTap) * xt.size(), h->s0); rt::copy_h2d(h->d_in_yt.p, yt.data(), sizeof(ResizeTap) * yt.size(), h->s0);
            rt::stream_sync(h->s0);
            h->in_tap_w = width; h->in_tap_h = height;
        }
        if (geom && C > 1 && h->d_frame.ensure((size_t)B * h->W * h->H * C + 16)) return fail(ORBX_E_DEVICE, "allocation failed");
    }
    const uint8_t* d_images = images;
    if (!on_device) {
        const size_t bytes = (size_t)(B - 1) * image_stride + (size_t)stride * (height - 1) + row_bytes;
        if (h->d_stage.ensure(bytes + 16)) return fail(ORBX_E_DEVICE, "staging allocation failed");
        if (rt::copy_h2d(h->d_stage.p, images, bytes, h->s0)) return fail(ORBX_E_DEVICE, "H2D copy failed: %s", rt::last_error());
        d_images = h->d_stage.p;
    }
    // input uploaded by orbx_device_upload_async: the extraction (eager or replayed) starts behind it
    if (h->copy_pending) {
        if (rt::stream_wait_event(h->s0, h->ev_copy)) return fail(ORBX_E_DEVICE, "extraction could not be ordered behind the input upload: %s", rt::last_error());
        h->copy_pending = false;
    }
#ifndef ORBX_EMU
    // hipGraph replay of the whole extraction (import, 7 dependent resize launches, FAST, quadtree, blur on the second stream,
    // layout, orient+BRIEF): at small batches the ~17 launches are launch/latency-bound.  The graph is keyed on everything that
    // is baked into the kernel arguments and re-captured when any of it changes.
    if (h->use_graph && !h->profile && !h->in_active) {
        const bool same = h->graph_exec && h->g_B == B && h->g_images == d_images && h->g_stride == stride && h->g_image_stride == image_stride &&
                          h->g_lap0 == lap0 && h->g_lap1 == lap1 && h->g_W == h->W && h->g_H == h->H && h->g_pyr == h->d_pyr.p && h->g_gauss == h->gauss_variant && h->g_undist_gen == h->undist_gen && h->g_pyramid_mode == h->pyramid_mode && h->g_small_forms == h->small_forms;
        if (!same) {
            if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
            if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
            if (hipStreamBeginCapture(h->s0, hipStreamCaptureModeThreadLocal) != hipSuccess) return fail(ORBX_E_DEVICE, "graph capture failed to start");
            rc = enqueue_extract(h, B, d_images, width, height, stride, image_stride, lap0, lap1);
            const hipError_t e = hipStreamEndCapture(h->s0, &h->graph);
            if (rc || e != hipSuccess || hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0) != hipSuccess) {
                h->graph_exec = nullptr;
                return fail(ORBX_E_DEVICE, "graph capture/instantiate failed: %s", rt::last_error());
            }
            h->g_B = B; h->g_images = d_images; h->g_stride = stride; h->g_image_stride = image_stride; h->g_lap0 = lap0; h->g_lap1 = lap1;
            h->g_W = h->W; h->g_H = h->H; h->g_pyr = h->d_pyr.p; h->g_gauss = h->gauss_variant; h->g_undist_gen = h->undist_gen; h->g_pyramid_mode = h->pyramid_mode; h->g_small_forms = h->small_forms;
        }
        if (hipGraphLaunch(h->graph_exec, h->s0) != hipSuccess) return fail(ORBX_E_DEVICE, "graph launch failed: %s", rt::last_error());
        // the records inside the capture belong to the graph; these are the ones other streams can wait on (an upload into the input buffer
        // waits for ev_import: after a replay that is the end of the whole graph, which is later than needed but never too early)
        h->import_lazy = true; h->done_lazy = true;
        h->lastB = B; h->extract_gen++; h->ex_undist_gen = h->undist_gen; h->ex_undist_active = h->undist.active != 0;
        return ORBX_OK;
    }
#endif
    return enqueue_extract(h, B, d_images, width, height, stride, image_stride, lap0, lap1);
}

int orbx_set_input(orbx_extractor* h, const OrbxInputSpec* spec) {
    if (!h) return fail(ORBX_E_ARG, "null handle");
    if (!spec) { h->in_active = false; return ORBX_OK; }
    if (spec->channels != 1 && spec->channels != 3 && spec->channels != 4) return fail(ORBX_E_ARG, "channels must be 1, 3 or 4");
    if (spec->geometry < 0 || spec->geometry > 2) return fail(ORBX_E_ARG, "unknown geometry");
    if (spec->geometry != 0 && (spec->out_w <= 0 || spec->out_h <= 0)) return fail(ORBX_E_ARG, "output size missing");
    if (spec->geometry == 1 && (!spec->map_x || !spec->map_y)) return fail(ORBX_E_ARG, "rectification maps missing");
    rt::set_device(h->device);
    if (spec->geometry == 1) {
        const size_t n = (size_t)spec->out_w * spec->out_h;
        if (h->d_mapx.ensure(n) || h->d_mapy.ensure(n)) return fail(ORBX_E_DEVICE, "allocation failed");
        if (rt::copy_h2d(h->d_mapx.p, spec->map_x, n * sizeof(float), h->s0) | rt::copy_h2d(h->d_mapy.p, spec->map_y, n * sizeof(float), h->s0) | rt::stream_sync(h->s0))
            return fail(ORBX_E_DEVICE, "map upload failed");
    }
    h->in_channels = spec->channels; h->in_rgb = spec->rgb ? 1 : 0; h->in_gray_variant = spec->gray_variant ? 1 : 0; h->in_geometry = spec->geometry;
    h->in_out_w = spec->out_w; h->in_out_h = spec->out_h; h->in_tap_w = h->in_tap_h = 0;
    h->in_active = true;
    return ORBX_OK;
}

int orbx_sync(orbx_extractor* h) {
    if (!h) return fail(ORBX_E_ARG, "null handle");
    rt::set_device(h->device);
    if (rt::stream_sync(h->s0) || rt::stream_sync(h->s1)) return fail(ORBX_E_DEVICE, "stream sync failed: %s", rt::last_error());
    if (h->copy_pending && rt::stream_sync(h->s_copy)) return fail(ORBX_E_DEVICE, "copy stream sync failed: %s", rt::last_error());   // uploads nobody has consumed yet
    if (h->profile) for (int i = 0; i < ORBX_NSTAGES; i++) if (i != ST_MATCH) h->stage_ms[i] = rt::event_elapsed_ms(h->ev_stage[i][0], h->ev_stage[i][1]);
    return ORBX_OK;
}

int orbx_fetch(orbx_extractor* h, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out, int* mono_out) {
    if (!h || h->lastB <= 0) return fail(ORBX_E_ARG, "nothing to fetch");
    rt::set_device(h->device);
    const int B = h->lastB; const size_t tc = (size_t)h->kp_total_cap;
    const size_t kb = (size_t)B * tc * sizeof(KeyPointRec), db = (size_t)B * tc * 32;
    const bool direct = (cap == h->kp_total_cap);   // caller's layout == device layout: no staging, no repack
    if (!direct && h->h_stage.ensure(kb + db + 64)) return fail(ORBX_E_DEVICE, "pinned allocation failed");
    int e = rt::copy_d2h(h->h_nm.p, h->d_nm.p, sizeof(int) * (2 * h->maxB + 1), h->s0);      // counts, mono indices and the status word behind them
    if (kps) e |= rt::copy_d2h(direct ? (void*)kps : (void*)h->h_stage.p, h->d_kps.p, kb, h->s0);
    if (desc) e |= rt::copy_d2h(direct ? (void*)desc : (void*)(h->h_stage.p + kb), h->d_desc.p, db, h->s0);
    if (e || rt::stream_sync(h->s0)) return fail(ORBX_E_DEVICE, "D2H failed: %s", rt::last_error());
    if (h->profile) orbx_sync(h);
    if (h->h_nm.p[2 * h->maxB] != 0) return fail(ORBX_E_INTERNAL, "device quadtree capacity check tripped");
    int rc = ORBX_OK;
    for (int b = 0; b < B; b++) {
        const int n = h->h_nm.p[b];
        if (n_out) n_out[b] = n;
        if (mono_out) mono_out[b] = h->h_nm.p[h->maxB + b];
        if (direct) continue;
        if (n > cap) { rc = ORBX_E_CAPACITY; continue; }
        if (kps) memcpy(kps + (size_t)b * cap, h->h_stage.p + (size_t)b * tc * sizeof(KeyPointRec), (size_t)n * sizeof(KeyPointRec));
        if (desc) memcpy(desc + (size_t)b * cap * 32, h->h_stage.p + kb + (size_t)b * tc * 32, (size_t)n * 32);
    }
    if (rc) return fail(rc, "output capacity %d too small", cap);
    return ORBX_OK;
}

int orbx_extract(orbx_extractor* h, const uint8_t* image, int width, int height, int stride, int lap0, int lap1,
                 OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out, int* mono_out) {
    if (n_out) *n_out = 0;
    if (mono_out) *mono_out = -1;
    int rc = orbx_extract_batch(h, 1, image, width, height, stride, (size_t)stride * height, 0, lap0, lap1);
    if (rc) return rc;
    return orbx_fetch(h, kps, desc, cap, n_out, mono_out);
}

int orbx_pyramid_level(orbx_extractor* h, int b, int level, int blurred, uint8_t* dst, int dst_stride, int* width, int* height) {
    if (!h || level < 0 || level >= h->nlevels || b < 0 || b >= h->lastB) return fail(ORBX_E_ARG, "bad pyramid query");
    const LevelInfo& L = h->lv[level];
    if (width) *width = L.w;
    if (height) *height = L.h;
    if (!dst) return ORBX_OK;
    if (dst_stride < L.w) return fail(ORBX_E_ARG, "dst_stride < width");
    rt::set_device(h->device);
    const size_t plane = (size_t)L.pitch * L.h;
    if (h->h_stage.ensure(plane + 64)) return fail(ORBX_E_DEVICE, "pinned allocation failed");
    rt::stream_sync(h->s1);
    const uint8_t* src = (blurred ? h->d_blur.p : h->d_pyr.p) + (size_t)b * h->pyr_stride + L.off;
    if (rt::copy_d2h(h->h_stage.p, src, plane, h->s0) || rt::stream_sync(h->s0)) return fail(ORBX_E_DEVICE, "D2H failed");
    for (int y = 0; y < L.h; y++) memcpy(dst + (size_t)y * dst_stride, h->h_stage.p + (size_t)y * L.pitch, L.w);
    return ORBX_OK;
}

// every level of one image in ONE device-to-host copy (the levels of an image are contiguous on the device): dst[l] / dst_stride[l] per level
int orbx_pyramid_fetch(orbx_extractor* h, int b, int blurred, uint8_t* const* dst, const int* dst_stride) {
    if (!h || !dst || !dst_stride || b < 0 || b >= h->lastB) return fail(ORBX_E_ARG, "bad pyramid query");
    for (int l = 0; l < h->nlevels; l++) if (!dst[l] || dst_stride[l] < h->lv[l].w) return fail(ORBX_E_ARG, "level %d: destination missing or too narrow", l);
    rt::set_device(h->device);
    if (h->h_stage.ensure(h->pyr_stride + 64)) return fail(ORBX_E_DEVICE, "pinned allocation failed");
    rt::stream_sync(h->s1);
    const uint8_t* src = (blurred ? h->d_blur.p : h->d_pyr.p) + (size_t)b * h->pyr_stride;
    if (rt::copy_d2h(h->h_stage.p, src, h->pyr_stride, h->s0) || rt::stream_sync(h->s0)) return fail(ORBX_E_DEVICE, "D2H failed");
    for (int l = 0; l < h->nlevels; l++) {
        const LevelInfo& L = h->lv[l];
        for (int y = 0; y < L.h; y++) memcpy(dst[l] + (size_t)y * dst_stride[l], h->h_stage.p + L.off + (size_t)y * L.pitch, L.w);
    }
    return ORBX_OK;
}

int orbx_device_alloc(orbx_extractor* h, size_t bytes, void** dptr) {
    if (!h || !dptr) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    *dptr = rt::dmalloc(bytes);
    return *dptr ? ORBX_OK : fail(ORBX_E_DEVICE, "hipMalloc(%zu) failed", bytes);
}
int orbx_device_free(orbx_extractor* h, void* dptr) { if (!h) return ORBX_E_ARG; rt::set_device(h->device); rt::dfree(dptr); return ORBX_OK; }
int orbx_device_upload(orbx_extractor* h, void* dptr, const void* host, size_t bytes) {
    if (!h || !dptr || !host) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    if (rt::copy_h2d(dptr, host, bytes, h->s0) || rt::stream_sync(h->s0)) return fail(ORBX_E_DEVICE, "upload failed");
    return ORBX_OK;
}

// Asynchronous input upload on the handle's copy stream: returns at once; the next orbx_extract_batch(on_device = 1) of this handle waits
// for it on the device.  With two device buffers a caller uploads batch i + 1 while batch i is being processed; an upload into the buffer
// the previous extraction read from waits (on the device) until that extraction has imported its level 0.
int orbx_device_upload_async(orbx_extractor* h, void* dptr, const void* host, size_t bytes) {
    if (!h || !dptr || !host) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    record_import_if_pending(h);
    if (h->lastB > 0 && rt::stream_wait_event(h->s_copy, h->ev_import)) return fail(ORBX_E_DEVICE, "upload could not be ordered behind the previous extraction: %s", rt::last_error());
    if (rt::copy_h2d(dptr, host, bytes, h->s_copy) || rt::event_record(h->ev_copy, h->s_copy)) return fail(ORBX_E_DEVICE, "upload failed: %s", rt::last_error());
    h->copy_pending = true;
    return ORBX_OK;
}

// debug: phase timestamps (100 MHz wall clock) of the level-0 quadtree workgroup of image 0 (serial profiling mode only)
int orbx_debug_quadtree_profile(orbx_extractor* h, long long out[16]) {
    if (!h || !out || !h->d_qtprof.p) return ORBX_E_ARG;
    rt::set_device(h->device);
    if (rt::copy_d2h(out, h->d_qtprof.p, sizeof(long long) * 16, h->s0) || rt::stream_sync(h->s0)) return ORBX_E_DEVICE;
    return ORBX_OK;
}

// debug: the instruction wrappers of csrc/orbx_simd.h applied to n operand triples; out: kSimdSelftestOps (22) x n results (see k_simd_selftest)
int orbx_debug_simd_selftest(orbx_extractor* h, const uint32_t* a, const uint32_t* b, const uint32_t* c, int n, uint32_t* out) {
    if (!h || !a || !b || !c || !out || n <= 0 || (n & 255)) return fail(ORBX_E_ARG, "bad arguments (n must be a positive multiple of 256)");
    rt::set_device(h->device);
    orbx::DevBuf<uint32_t> d;
    const size_t N = (size_t)n;
    if (d.ensure((3 + kSimdSelftestOps) * N)) return fail(ORBX_E_DEVICE, "allocation failed");
    int e = rt::copy_h2d(d.p, a, 4 * N, h->s0) | rt::copy_h2d(d.p + N, b, 4 * N, h->s0) | rt::copy_h2d(d.p + 2 * N, c, 4 * N, h->s0);
    dim3 grid((n + 255) / 256, 1, 1), blk(256, 1, 1);
    ORBX_LAUNCH(k_simd_selftest, grid, blk, 0, h->s0, (const uint32_t*)d.p, (const uint32_t*)(d.p + N), (const uint32_t*)(d.p + 2 * N), n, d.p + 3 * N);
    e |= rt::copy_d2h(out, d.p + 3 * N, 4 * N * kSimdSelftestOps, h->s0);
    if (e || rt::stream_sync(h->s0) || rt::check_launch()) { d.release(); return fail(ORBX_E_DEVICE, "self-test failed: %s", rt::last_error()); }
    d.release();
    return ORBX_OK;
}

int orbx_debug_live_resources(long long out[4]) {
    if (!out) return fail(ORBX_E_ARG, "null out");
    const rt::Live& l = rt::live();
    out[0] = l.dev.load(); out[1] = l.pinned.load(); out[2] = l.streams.load(); out[3] = l.events.load();
    return ORBX_OK;
}

int orbx_host_alloc(orbx_extractor* h, size_t bytes, void** hptr) {
    if (!h || !hptr) return fail(ORBX_E_ARG, "null");
    rt::set_device(h->device);
    *hptr = rt::hmalloc(bytes);
    return *hptr ? ORBX_OK : fail(ORBX_E_DEVICE, "pinned host allocation of %zu bytes failed", bytes);
}
int orbx_host_free(orbx_extractor* h, void* hptr) { if (!h) return ORBX_E_ARG; rt::set_device(h->device); rt::hfree(hptr); return ORBX_OK; }

int orbx_debug_stereo_flags(orbx_extractor* h, int flags) { if (!h) return ORBX_E_ARG; h->debug_stereo_flags = flags; return ORBX_OK; }

// device addresses of the results of the last batch (layout [B][cap]...; valid until the handle is reconfigured or destroyed), for consumers
// that stay on the device: RCCL collectives over the descriptor blocks, a global matcher
int orbx_device_outputs(orbx_extractor* h, void** kps, void** desc, void** n, void** mono, int* cap, int* B) {
    if (!h || h->lastB <= 0) return fail(ORBX_E_ARG, "nothing extracted yet");
    if (kps) *kps = h->d_kps.p;
    if (desc) *desc = h->d_desc.p;
    if (n) *n = h->d_nm.p;
    if (mono) *mono = h->d_nm.p + h->maxB;
    if (cap) *cap = h->kp_total_cap;
    if (B) *B = h->lastB;
    return ORBX_OK;
}

int orbx_set_undistort(orbx_extractor* h, const float K[4], const float* dist, int ndist, int opencv_variant) {
    if (!h) return fail(ORBX_E_ARG, "null handle");
    h->undist_gen++;
    memset(&h->undist, 0, sizeof h->undist);
    if (!K || !dist || ndist <= 0 || dist[0] == 0.0f) return ORBX_OK;           // mDistCoef.at<float>(0) == 0.0: mvKeysUn = mvKeys (src/Frame.cc:1005-1009)
    if (ndist != 4 && ndist != 5) return fail(ORBX_E_ARG, "distortion coefficients: k1, k2, p1, p2[, k3]");
    if (opencv_variant != 0 && opencv_variant != 1) return fail(ORBX_E_ARG, "bad OpenCV variant");
    UndistortParams& U = h->undist;
    U.fx = K[0]; U.fy = K[1]; U.cx = K[2]; U.cy = K[3]; U.ifx = 1. / U.fx; U.ify = 1. / U.fy;
    for (int i = 0; i < ndist; i++) U.k[i] = dist[i];
    U.variant = opencv_variant; U.active = 1;
    return ORBX_OK;
}
int orbx_fetch_undistorted(orbx_extractor* h, OrbxKeyPoint* kps_un, int cap) {
    if (!h || !kps_un || h->lastB <= 0) return fail(ORBX_E_ARG, "nothing to fetch");
    if (cap < h->kp_total_cap) return fail(ORBX_E_CAPACITY, "rows need %d entries", h->kp_total_cap);
    rt::set_device(h->device);
    const size_t tc = (size_t)h->kp_total_cap;
    if (undistort_stale(h)) return fail(ORBX_E_ARG, "orbx_set_undistort was called after the last extraction: mvKeysUn of that batch belongs to the previous model - extract again");
    const KeyPointRec* src = h->ex_undist_active ? h->d_kps_un.p : h->d_kps.p;      // without distortion mvKeysUn is mvKeys
    int e = 0;
    if ((size_t)cap == tc) e = rt::copy_d2h(kps_un, src, (size_t)h->lastB * tc * sizeof(KeyPointRec), h->s0);
    else for (int b = 0; b < h->lastB; b++) e |= rt::copy_d2h(kps_un + (size_t)b * cap, src + (size_t)b * tc, tc * sizeof(KeyPointRec), h->s0);
    if (e || rt::stream_sync(h->s0)) return fail(ORBX_E_DEVICE, "D2H failed: %s", rt::last_error());
    return ORBX_OK;
}
int orbx_undistorted_bounds(const orbx_extractor* h, int width, int height, float out[4]) {
    if (!h || !out) return fail(ORBX_E_ARG, "null");
    if (!h->undist.active) { out[0] = 0.0f; out[1] = (float)width; out[2] = 0.0f; out[3] = (float)height; return ORBX_OK; }     // src/Frame.cc:1068-1074
    float x[4], y[4];
    const float cu[4] = {0.0f, (float)width, 0.0f, (float)width}, cv[4] = {0.0f, 0.0f, (float)height, (float)height};
    for (int i = 0; i < 4; i++) undistort_point(h->undist, cu[i], cv[i], &x[i], &y[i]);
    out[0] = std::min(x[0], x[2]); out[1] = std::max(x[1], x[3]); out[2] = std::min(y[0], y[1]); out[3] = std::max(y[2], y[3]);              // :1062-1065
    return ORBX_OK;
}

int orbx_input_buffer(orbx_extractor* h, int width, int height, int B, void** dptr, int* stride, size_t* image_stride) {
    if (!h || !dptr) return fail(ORBX_E_ARG, "null");
    if (h->in_active && (h->in_channels != 1 || h->in_geometry != 0)) return fail(ORBX_E_ARG, "zero-copy input needs plain 8-bit grey frames (an input pre-step is set)");
    const int rc = configure(h, width, height, B); if (rc) return rc;
    *dptr = h->d_pyr.p + h->lv[0].off;
    if (stride) *stride = h->lv[0].pitch;
    if (image_stride) *image_stride = h->pyr_stride;
    return ORBX_OK;
}
int orbx_input_upload(orbx_extractor* h, int B, const uint8_t* images, int width, int height, int stride, size_t image_stride) {
    void* d = nullptr; int pitch = 0; size_t istr = 0;
    if (!images || stride < width) return fail(ORBX_E_ARG, "bad input images");
    int rc = orbx_input_buffer(h, width, height, B, &d, &pitch, &istr); if (rc) return rc;
    rt::set_device(h->device);
    int e = 0;
    for (int b = 0; b < B; b++) e |= rt::copy2d((uint8_t*)d + (size_t)b * istr, (size_t)pitch, images + (size_t)b * image_stride, (size_t)stride, (size_t)width, (size_t)height, 0, h->s0);
    if (e || rt::stream_sync(h->s0)) return fail(ORBX_E_DEVICE, "input upload failed: %s", rt::last_error());
    return ORBX_OK;
}

int orbx_device_snapshot(orbx_extractor* h, void* desc_dst, void* n_dst) {
    if (!h || h->lastB <= 0) return fail(ORBX_E_ARG, "nothing extracted yet");
    rt::set_device(h->device);
    int e = 0;
    if (desc_dst) e |= rt::copy_d2d(desc_dst, h->d_desc.p, (size_t)h->lastB * h->kp_total_cap * 32, h->s0);
    if (n_dst) e |= rt::copy_d2d(n_dst, h->d_nm.p, sizeof(int) * (size_t)h->lastB, h->s0);
    if (e || rt::stream_sync(h->s0)) return fail(ORBX_E_DEVICE, "snapshot failed: %s", rt::last_error());
    return ORBX_OK;
}
int orbx_device_id(const orbx_extractor* h) { return h ? (rt::memory_is_host() ? ORBX_DEVICE_HOST : h->device) : ORBX_E_ARG; }

int orbx_set_pyramid_mode(orbx_extractor* h, int mode) {
    if (!h || mode < 0 || mode > 2) return fail(ORBX_E_ARG, "pyramid mode 0 (by batch size), 1 (one launch per level) or 2 (one launch)");
    h->pyramid_mode = mode;
    return ORBX_OK;
}

int orbx_set_small_batch_forms(orbx_extractor* h, int on) { if (!h) return ORBX_E_ARG; h->small_forms = on != 0; return ORBX_OK; }
int orbx_set_host_wait(int device_id, int mode) {
    if (mode < 0 || mode > 3) return fail(ORBX_E_ARG, "host wait mode %d (0 default, 1 blocking, 2 spin, 3 yield)", mode);
    if (device_id < 0 || device_id >= rt::device_count()) return fail(ORBX_E_DEVICE, "no usable GPU %d", device_id);
    if (rt::set_host_wait(device_id, mode)) return fail(ORBX_E_DEVICE, "hipSetDeviceFlags failed: %s", rt::last_error());
    return ORBX_OK;
}
int orbx_debug_quadtree_pool_levels(orbx_extractor* h) { return h ? h->qt_pool_levels : ORBX_E_ARG; }
int orbx_debug_quadtree_lds_nodes(orbx_extractor* h, int max_nodes) {
    if (!h || max_nodes < 0) return ORBX_E_ARG;
    h->qt_lds_nodes = std::min(max_nodes, kQuadtreeLdsNodes);
    h->g_B = 0;                 // a captured graph holds the launches of the old plan
    return ORBX_OK;
}

int orbx_set_graph_replay(orbx_extractor* h, int on) { if (!h) return ORBX_E_ARG; h->use_graph = on != 0; return ORBX_OK; }

int orbx_profile_enable(orbx_extractor* h, int on) { if (!h) return ORBX_E_ARG; h->profile = on != 0; h->serial = on == 2; return ORBX_OK; }
int orbx_profile_get(orbx_extractor* h, float ms[ORBX_NSTAGES]) {
    if (!h || !ms) return ORBX_E_ARG;
    for (int i = 0; i < ORBX_NSTAGES; i++) ms[i] = h->stage_ms[i];
    return ORBX_OK;
}

// ---- stage probes -------------------------------------------------------------------------------------
int orbx_debug_candidates(orbx_extractor* h, int b, int level, int* xys, int cap) {
    if (!h || (!xys && cap > 0) || level < 0 || level >= h->nlevels || b < 0 || b >= h->lastB) return fail(ORBX_E_ARG, "bad probe");
    rt::set_device(h->device);
    rt::stream_sync(h->s0);
    const LevelInfo& L = h->lv[level];
    std::vector<int> counts(L.cell_count);
    rt::copy_d2h(counts.data(), h->d_cell_count.p + (size_t)b * h->ncells + L.cell_begin, sizeof(int) * L.cell_count, h->s0);
    std::vector<uint32_t> slots(L.cand_cap + 1);
    rt::copy_d2h(slots.data(), h->d_slots.p + (size_t)b * h->cand_stride + L.cand_off, sizeof(uint32_t) * L.cand_cap, h->s0);
    rt::stream_sync(h->s0);
    int n = 0;
    for (int c = 0; c < L.cell_count; c++) {
        const int so = h->cells[L.cell_begin + c].slot_off - L.cand_off;
        for (int k = 0; k < counts[c]; k++) {
            if (n < cap) { const uint32_t key = slots[so + k]; xys[3 * n] = key_x(key); xys[3 * n + 1] = key_y(key); xys[3 * n + 2] = key_s(key); }
            n++;
        }
    }
    return n;
}
int orbx_debug_level_keys(orbx_extractor* h, int b, int level, int* xys, int cap) {
    if (!h || (!xys && cap > 0) || level < 0 || level >= h->nlevels || b < 0 || b >= h->lastB) return fail(ORBX_E_ARG, "bad probe");
    rt::set_device(h->device);
    rt::stream_sync(h->s0);
    const LevelInfo& L = h->lv[level];
    int cnt = 0;
    rt::copy_d2h(&cnt, h->d_lvl_count.p + (size_t)b * h->nlevels + level, sizeof(int), h->s0);
    std::vector<uint32_t> keys(L.kp_cap);
    rt::copy_d2h(keys.data(), h->d_lvl_keys.p + (size_t)b * h->kp_total_cap + L.kp_off, sizeof(uint32_t) * L.kp_cap, h->s0);
    rt::stream_sync(h->s0);
    for (int i = 0; i < cnt && i < cap; i++) { xys[3 * i] = key_x(keys[i]); xys[3 * i + 1] = key_y(keys[i]); xys[3 * i + 2] = key_s(keys[i]); }
    return cnt;
}

// ---- matchers ------------------------------------------------------------------------------------------
int orbm_hamming_matrix(orbx_extractor* h, const uint8_t* a, int na, const uint8_t* b, int nb, int* out) {
    if (!h || !a || !b || !out || na <= 0 || nb <= 0) return fail(ORBX_E_ARG, "bad hamming arguments");
    rt::set_device(h->device);
    if (h->d_hamA.ensure((size_t)na * 4) || h->d_hamB.ensure((size_t)nb * 4) || h->d_hamOut.ensure((size_t)na * nb)) return fail(ORBX_E_DEVICE, "allocation failed");
    // both descriptor sets go up from page-locked memory of the handle, the matrix comes back into it (fetch_sync): the caller's arrays are pageable
    const size_t ba = (size_t)na * 32, bb = (size_t)nb * 32;
    if (h->h_packA.ensure(ba + bb + 16)) return fail(ORBX_E_DEVICE, "allocation failed");
    memcpy(h->h_packA.p, a, ba); memcpy(h->h_packA.p + ba, b, bb);
    rt::copy_h2d(h->d_hamA.p, h->h_packA.p, ba, h->s0);
    rt::copy_h2d(h->d_hamB.p, h->h_packA.p + ba, bb, h->s0);
    dim3 grid((nb + 255) / 256, na, 1), blk(256, 1, 1);
    ORBX_LAUNCH(k_hamming_matrix, grid, blk, 0, h->s0, (const unsigned long long*)h->d_hamA.p, na, (const unsigned long long*)h->d_hamB.p, nb, h->d_hamOut.p);
    if (fetch_sync(h, out, h->d_hamOut.p, sizeof(int) * (size_t)na * nb)) return fail(ORBX_E_DEVICE, "hamming failed: %s", rt::last_error());
    return ORBX_OK;
}

static int check_pair(orbx_extractor* L, int lf, orbx_extractor* R, int rf, int B) {
    if (!L || !R || B <= 0 || lf < 0 || rf < 0) return fail(ORBX_E_ARG, "bad matcher arguments");
    if (lf + B > L->lastB || rf + B > R->lastB) return fail(ORBX_E_ARG, "matcher range exceeds the last extracted batch");
    if (L->W != R->W || L->H != R->H || L->nlevels != R->nlevels || L->kp_total_cap != R->kp_total_cap || L->device != R->device)
        return fail(ORBX_E_ARG, "left/right extractors differ in geometry, parameters or device");
    return ORBX_OK;
}

int orbm_stereo_match(orbx_extractor* L, int lf, orbx_extractor* R, int rf, int B, float bf, float bl) {
    int rc = check_pair(L, lf, R, rf, B); if (rc) return rc;
    rt::set_device(L->device);
    if (L != R) { record_done_if_pending(R); rt::stream_wait_event(L->s0, R->ev_done); }
    const int cap = L->kp_total_cap;
    StereoParams P; P.mbf = bf; P.mb = bl; P.th_high = 100; P.th_orb = (100 + 50) / 2;   // ORBmatcher::TH_HIGH/TH_LOW, src/ORBmatcher.cc:35-36
    P.debug_flags = L->debug_stereo_flags;
    if (L->profile) rt::event_record(L->ev_stage[ST_MATCH][0], L->s0);
    dim3 grid((cap + 3) / 4, B, 1), blk(256, 1, 1);
    // row index of the right keypoints (buckets of 1 << kStereoRowShift rows, by the first row of each candidate band): built by the
    // right extractor's k_layout for every image of its batch
    const int nb = (R->H >> kStereoRowShift) + 2;
    const int lookback = (int)std::ceil(4.0f * R->scale[R->nlevels - 1]) + 2;     // tallest band: 2 * (2 * scale) + rounding
    ORBX_LAUNCH(k_stereo_match, grid, blk, 0, L->s0, (const LevelInfo*)L->d_lv.p,
                (const KeyPointRec*)(L->d_kps.p + (size_t)lf * cap), (const unsigned long long*)(L->d_desc.p + (size_t)lf * cap * 4), (const int*)(L->d_nm.p + lf),
                (const KeyPointRec*)(R->d_kps.p + (size_t)rf * cap), (const unsigned long long*)(R->d_desc.p + (size_t)rf * cap * 4),
                (const int4*)(R->d_aux.p + (size_t)rf * cap * 4), (const int*)(R->d_nm.p + rf),
                (const int*)(R->d_rowstart.p + (size_t)rf * (nb + 1)), (const int*)(R->d_rowitems.p + (size_t)rf * cap), nb, lookback, cap, (const uint8_t*)(L->d_pyr.p + (size_t)lf * L->pyr_stride), (const uint8_t*)(R->d_pyr.p + (size_t)rf * R->pyr_stride), L->pyr_stride,
                P, L->d_uRight.p, L->d_depth.p, L->d_sad.p);
    dim3 grid2(B, 1, 1);
    ORBX_LAUNCH(k_stereo_median, grid2, blk, 0, L->s0, (const int*)(L->d_nm.p + lf), cap, L->d_uRight.p, L->d_depth.p,
                (const int*)L->d_sad.p, L->d_nmatch.p);
    if (L->profile) rt::event_record(L->ev_stage[ST_MATCH][1], L->s0);
    if (rt::check_launch()) return fail(ORBX_E_DEVICE, "kernel launch failed: %s", rt::last_error());
    return ORBX_OK;
}

int orbm_stereo_fetch(orbx_extractor* L, int B, float* uRight, float* depth, int cap, int* n_matches) {
    if (!L || B <= 0 || B > L->maxB) return fail(ORBX_E_ARG, "bad fetch");
    rt::set_device(L->device);
    const size_t tc = (size_t)L->kp_total_cap;
    const bool direct = ((size_t)cap == tc);
    if (!direct && L->h_stage.ensure(2 * B * tc * sizeof(float) + 64)) return fail(ORBX_E_DEVICE, "pinned allocation failed");
    float* hu = direct ? uRight : (float*)L->h_stage.p; float* hd = direct ? depth : hu + B * tc;
    int e = 0;
    if (hu) e |= rt::copy_d2h(hu, L->d_uRight.p, B * tc * sizeof(float), L->s0);
    if (hd) e |= rt::copy_d2h(hd, L->d_depth.p, B * tc * sizeof(float), L->s0);
    e |= rt::copy_d2h(L->h_nm.p, L->d_nmatch.p, sizeof(int) * B, L->s0);
    if (e || rt::stream_sync(L->s0)) return fail(ORBX_E_DEVICE, "D2H failed: %s", rt::last_error());
    if (L->profile) L->stage_ms[ST_MATCH] = rt::event_elapsed_ms(L->ev_stage[ST_MATCH][0], L->ev_stage[ST_MATCH][1]);
    const size_t ncopy = std::min<size_t>(tc, (size_t)cap);
    for (int b = 0; b < B; b++) {
        if (!direct && uRight) memcpy(uRight + (size_t)b * cap, hu + b * tc, ncopy * sizeof(float));
        if (!direct && depth) memcpy(depth + (size_t)b * cap, hd + b * tc, ncopy * sizeof(float));
        if (n_matches) n_matches[b] = L->h_nm.p[b];
    }
    return ORBX_OK;
}

int orbm_knn2(orbx_extractor* L, int lf, orbx_extractor* R, int rf, int B) {
    int rc = check_pair(L, lf, R, rf, B); if (rc) return rc;
    rt::set_device(L->device);
    if (L != R) { record_done_if_pending(R); rt::stream_wait_event(L->s0, R->ev_done); }
    const int cap = L->kp_total_cap; const size_t bc = (size_t)L->maxB * cap;
    if (L->profile) rt::event_record(L->ev_stage[ST_MATCH][0], L->s0);
    const dim3 blk(256, 1, 1);
    const int* qoff = L->d_nm.p + L->maxB + lf; const int* nq = L->d_nm.p + lf;
    const int* toff = R->d_nm.p + R->maxB + rf; const int* nt = R->d_nm.p + rf;
    if (L->debug_stereo_flags & 8) {                       // test switch: the wave-per-query kernel on the vector units
        dim3 grid((cap + 3) / 4, B, 1);
        ORBX_LAUNCH(k_knn2, grid, blk, 0, L->s0,
                    (const unsigned long long*)(L->d_desc.p + (size_t)lf * cap * 4), qoff, nq,
                    (const unsigned long long*)(R->d_desc.p + (size_t)rf * cap * 4), toff, nt,
                    cap, L->d_knn.p, L->d_knn.p + bc, L->d_knn.p + 2 * bc, L->d_knn.p + 3 * bc, L->d_ratio.p);
    } else {
        // the distance matrix on the matrix cores with the top-2 in its epilogue
        dim3 grid((cap + 31) / 32, B, 1);
        ORBX_LAUNCH(k_knn2_mfma, grid, blk, 0, L->s0,
                    (const unsigned long long*)(L->d_desc.p + (size_t)lf * cap * 4), qoff, nq,
                    (const unsigned long long*)(R->d_desc.p + (size_t)rf * cap * 4), toff, nt,
                    cap, L->d_knn.p, L->d_knn.p + bc, L->d_knn.p + 2 * bc, L->d_knn.p + 3 * bc, L->d_ratio.p);
    }
    if (L->profile) rt::event_record(L->ev_stage[ST_MATCH][1], L->s0);
    if (rt::check_launch()) return fail(ORBX_E_DEVICE, "kernel launch failed: %s", rt::last_error());
    return ORBX_OK;
}

// Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1530-1587) for B fisheye pairs: 2-NN + ratio test (k_knn2), then the triangulation gate
// KannalaBrandt8::TriangulateMatches (src/CameraModels/KannalaBrandt8.cpp:439-523) on the device.
int orbm_stereo_fisheye(orbx_extractor* L, int lf, orbx_extractor* R, int rf, int B, const OrbmKB8Stereo* S) {
    if (!S) return fail(ORBX_E_ARG, "null camera parameters");
    int rc = orbm_knn2(L, lf, R, rf, B); if (rc) return rc;
    const int cap = L->kp_total_cap; const size_t bc = (size_t)L->maxB * cap;
    if (L->d_l2r.ensure(bc) || L->d_r2l.ensure(bc) || L->d_p3d.ensure(3 * bc)) return fail(ORBX_E_DEVICE, "allocation failed");
    KB8StereoParams P; memset(&P, 0, sizeof P);
    for (int i = 0; i < 8; i++) { P.cam1[i] = S->cam1[i]; P.cam2[i] = S->cam2[i]; }
    for (int i = 0; i < 9; i++) P.R12[i] = S->R12[i];
    for (int i = 0; i < 3; i++) P.t12[i] = S->t12[i];
    for (int l = 0; l < L->nlevels; l++) P.sigma2[l] = L->sigma2[l];
    rt::memset_async(L->d_r2l.p, 0xFF, sizeof(int) * (size_t)B * cap, L->s0);
    rt::memset_async(L->d_nmatch.p, 0, sizeof(int) * (size_t)B, L->s0);
    dim3 grid((cap + 255) / 256, B, 1), blk(256, 1, 1);
    ORBX_LAUNCH(k_kb8_stereo, grid, blk, 0, L->s0, (const KeyPointRec*)(L->d_kps.p + (size_t)lf * cap), (const int*)(L->d_nm.p + L->maxB + lf), (const int*)(L->d_nm.p + lf),
                (const KeyPointRec*)(R->d_kps.p + (size_t)rf * cap), (const int*)(R->d_nm.p + R->maxB + rf), cap, (const int*)L->d_knn.p, (const uint8_t*)L->d_ratio.p, P,
                L->d_l2r.p, L->d_r2l.p, L->d_depth.p, L->d_p3d.p, L->d_nmatch.p);
    if (L->profile) rt::event_record(L->ev_stage[ST_MATCH][1], L->s0);
    if (rt::check_launch()) return fail(ORBX_E_DEVICE, "kernel launch failed: %s", rt::last_error());
    return ORBX_OK;
}

int orbm_stereo_fisheye_fetch(orbx_extractor* L, int B, int* l2r, int* r2l, float* depth, float* p3d, int* n_matches, int cap) {
    if (!L || B <= 0 || B > L->maxB || !L->d_l2r.p) return fail(ORBX_E_ARG, "bad fetch");
    rt::set_device(L->device);
    const size_t tc = (size_t)L->kp_total_cap, n = (size_t)B * tc;
    if (L->h_stage.ensure(n * (2 * sizeof(int) + 4 * sizeof(float)) + 64)) return fail(ORBX_E_DEVICE, "pinned allocation failed");
    int* hl = (int*)L->h_stage.p; int* hr = hl + n; float* hd = (float*)(hr + n); float* hp = hd + n;
    int e = rt::copy_d2h(hl, L->d_l2r.p, n * sizeof(int), L->s0) | rt::copy_d2h(hr, L->d_r2l.p, n * sizeof(int), L->s0) |
            rt::copy_d2h(hd, L->d_depth.p, n * sizeof(float), L->s0) | rt::copy_d2h(hp, L->d_p3d.p, 3 * n * sizeof(float), L->s0) |
            rt::copy_d2h(L->h_nm.p, L->d_nmatch.p, sizeof(int) * B, L->s0);
    if (e || rt::stream_sync(L->s0)) return fail(ORBX_E_DEVICE, "D2H failed: %s", rt::last_error());
    if (L->profile) L->stage_ms[ST_MATCH] = rt::event_elapsed_ms(L->ev_stage[ST_MATCH][0], L->ev_stage[ST_MATCH][1]);
    const size_t ncopy = std::min<size_t>(tc, (size_t)cap);
    for (int b = 0; b < B; b++) {
        if (l2r) memcpy(l2r + (size_t)b * cap, hl + b * tc, ncopy * sizeof(int));
        if (r2l) memcpy(r2l + (size_t)b * cap, hr + b * tc, ncopy * sizeof(int));
        if (depth) memcpy(depth + (size_t)b * cap, hd + b * tc, ncopy * sizeof(float));
        if (p3d) memcpy(p3d + 3 * (size_t)b * cap, hp + 3 * b * tc, 3 * ncopy * sizeof(float));
        if (n_matches) n_matches[b] = L->h_nm.p[b];
    }
    return ORBX_OK;
}

int orbm_knn2_fetch(orbx_extractor* L, int B, int* idx0, int* dist0, int* idx1, int* dist1, uint8_t* ratio_ok, int cap) {
    if (!L || B <= 0 || B > L->maxB) return fail(ORBX_E_ARG, "bad fetch");
    rt::set_device(L->device);
    const size_t tc = (size_t)L->kp_total_cap, bc = (size_t)L->maxB * tc;
    if (L->h_stage.ensure(4 * B * tc * sizeof(int) + B * tc + 64)) return fail(ORBX_E_DEVICE, "pinned allocation failed");
    int* hk = (int*)L->h_stage.p; uint8_t* hr = L->h_stage.p + 4 * B * tc * sizeof(int);
    int e = 0;
    for (int k = 0; k < 4; k++) e |= rt::copy_d2h(hk + k * B * tc, L->d_knn.p + k * bc, B * tc * sizeof(int), L->s0);
    e |= rt::copy_d2h(hr, L->d_ratio.p, B * tc, L->s0);
    if (e || rt::stream_sync(L->s0)) return fail(ORBX_E_DEVICE, "D2H failed: %s", rt::last_error());
    if (L->profile) L->stage_ms[ST_MATCH] = rt::event_elapsed_ms(L->ev_stage[ST_MATCH][0], L->ev_stage[ST_MATCH][1]);
    const size_t ncopy = std::min<size_t>(tc, (size_t)cap);
    int* outs[4] = {idx0, dist0, idx1, dist1};
    for (int b = 0; b < B; b++) {
        for (int k = 0; k < 4; k++) if (outs[k]) memcpy(outs[k] + (size_t)b * cap, hk + k * B * tc + b * tc, ncopy * sizeof(int));
        if (ratio_ok) memcpy(ratio_ok + (size_t)b * cap, hr + b * tc, ncopy);
    }
    return ORBX_OK;
}

}  // extern "C"
