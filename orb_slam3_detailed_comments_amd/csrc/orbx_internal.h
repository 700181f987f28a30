// orbx_internal.h — host-side internals shared by orbx_api.cpp (extractor + stereo) and orbm_search.cpp
// (projection / BoW searches): error reporting, RAII-less device/pinned buffers, the handle struct.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/orbx.h"
#include "orbx_kernels.h"
#include "orbx_rt.h"

namespace orbx {

int fail(int code, const char* fmt, ...);

inline int round_half_even_f(float v) { return (int)lrintf(v); }    // cvRound under the default FP mode
inline int round_half_even_d(double v) { return (int)lrint(v); }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

enum Stage { ST_IMPORT = 0, ST_PYRAMID, ST_FAST, ST_QUADTREE, ST_BLUR, ST_LAYOUT, ST_DESCRIBE, ST_MATCH };

template <typename T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    int ensure(size_t count) {
        if (count <= n && p) return 0;
        rt::dfree(p); p = (T*)rt::dmalloc(count * sizeof(T)); n = p ? count : 0;
        return p ? 0 : -1;
    }
    void release() { rt::dfree(p); p = nullptr; n = 0; }
};
template <typename T> struct HostBuf {
    T* p = nullptr; size_t n = 0;
    int ensure(size_t count) {
        if (count <= n && p) return 0;
        rt::hfree(p); p = (T*)rt::hmalloc(count * sizeof(T)); n = p ? count : 0;
        return p ? 0 : -1;
    }
    void release() { rt::hfree(p); p = nullptr; n = 0; }
};

// where the vocabulary transform of the last orbv_transform_extracted left the FeatureVectors of its frames (orbv_api.cpp), for the searches that
// read them in place (orbm_search_by_bow_frames_batch): frame b's sorted node ids at fv_node + b * cap, CSR offsets at fv_start + b * (cap + 1),
// feature indices at fv_feat + b * cap, its node count at nout[2 * b + 1]
struct VocFrameArrays { const uint32_t* fv_node; const int* fv_start; const int* fv_feat; const int* nout; int cap, lastB, device, first; const void* handle; uint64_t extract_gen; };
}  // namespace orbx
struct orbv_vocabulary;
namespace orbx { int orbv_frame_arrays(const orbv_vocabulary* v, VocFrameArrays* out); }

struct orbx_extractor {
    // ---- reference constructor state (src/ORBextractor.cc:468-571) ----
    int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0, device = 0, gauss_variant = 0;
    double scaleFactor = 1.0;   // the reference keeps the float argument in a double member (include/ORBextractor.h:96)
    float scale[orbx::kMaxLevels], inv_scale[orbx::kMaxLevels], sigma2[orbx::kMaxLevels], inv_sigma2[orbx::kMaxLevels];
    int quota[orbx::kMaxLevels];
    orbx::UmaxTab umax;
    // ---- geometry for the configured resolution ----
    int W = 0, H = 0, maxB = 0;
    orbx::LevelInfo lv[orbx::kMaxLevels];
    std::vector<orbx::CellInfo> cells;
    std::vector<orbx::ResizeTap> xtab, ytab;
    bool resize_rows_ok[orbx::kMaxLevels] = {};     // level l can use k_resize_rows (scale factor <= 2)
    // k_pyramid_fused (small batches: all levels in one launch): tiles of the top level, their regions / owned intervals per level
    std::vector<orbx::PyrSpan> xspan, yspan;
    int pyr_ntx = 0, pyr_nty = 0, pyr_buf_a = 0, pyr_buf_b = 0;
    orbx::PyrTapOffsets pyr_toff = {};
    bool pyr_fused_ok = false;
    bool small_forms = true, g_small_forms = true;  // orbx_set_small_batch_forms: blur + FAST in one launch on one stream (batches <= 32)
    int pyramid_mode = 0, g_pyramid_mode = 0;       // orbx_set_pyramid_mode: 0 = by batch size, 1 = one launch per level, 2 = one launch
    size_t pyr_stride = 0, cand_stride = 0;
    int ncells = 0, kp_total_cap = 0, node_cap = 0, nb_cap = 1, fast_tile_bytes = 0, fast_inner_bytes = 0;
    // ---- device state ----
    orbx::DevBuf<orbx::LevelInfo> d_lv; orbx::DevBuf<orbx::CellInfo> d_cells; orbx::DevBuf<orbx::ResizeTap> d_xtab, d_ytab;
    orbx::DevBuf<orbx::PyrSpan> d_xspan, d_yspan;
    orbx::DevBuf<uint8_t> d_pyr, d_blur, d_stage;
    orbx::DevBuf<uint32_t> d_slots, d_candA, d_candB, d_lvl_keys;
    orbx::DevBuf<int> d_cell_count, d_lvl_count, d_final_idx, d_nm, d_status;
    orbx::DevBuf<orbx::KeyPointRec> d_kps; orbx::DevBuf<unsigned long long> d_desc;
    orbx::DevBuf<float> d_uRight, d_depth; orbx::DevBuf<int> d_sad, d_nmatch;
    orbx::DevBuf<int> d_knn; orbx::DevBuf<uint8_t> d_ratio;
    orbx::DevBuf<int> d_l2r, d_r2l; orbx::DevBuf<float> d_p3d;     // fisheye stereo (orbm_stereo_fisheye)
    orbx::DevBuf<unsigned long long> d_hamA, d_hamB; orbx::DevBuf<int> d_hamOut;
    orbx::HostBuf<uint8_t> h_stage;
    orbx::HostBuf<int> h_nm;
    orbx::rt::stream_t s0 = 0, s1 = 0, s_copy = 0;          // s_copy: orbx_device_upload_async (input uploads beside the kernels of the previous batch)
    orbx::rt::event_t ev_fork = 0, ev_join = 0, ev_done = 0, ev_copy = 0, ev_import = 0;
    bool copy_pending = false;
    // ev_done / ev_import are recorded when somebody is about to wait for them (another handle's stereo search, an input upload on s_copy), not
    // after every extraction: a record in the middle of a stream is a barrier packet, and the kernel behind it starts ~6 us later - a third of a
    // FAST launch at one pair per call.  Recorded late they cover more of the stream than needed, never less.
    bool done_lazy = false, import_lazy = false;
    orbx::rt::event_t ev_stage[ORBX_NSTAGES][2] = {};
    bool profile = false, serial = false, have_streams = false;
    int lastB = 0;
    uint64_t extract_gen = 0;     // counts the extractions enqueued on this handle: results derived from a batch (the vocabulary transform's FeatureVectors) name the one they belong to
    int debug_stereo_flags = 0;   // orbx_debug_stereo_flags (tests): bit 0 reversed candidate visiting order, bit 1 round-1 distance-only compare, bit 4 matrix form of Tcw * p (sophus_action.h)
    float stage_ms[ORBX_NSTAGES];
    // scratch of the projection / BoW searches (orbm_search.cpp)
    orbx::DevBuf<uint8_t> d_sr[12];
    orbx::HostBuf<uint8_t> h_packA, h_packB, h_out;      // pinned staging of the window searches (frame, queries, results)
    orbx::HostBuf<uint8_t> h_res;                        // pinned landing area of results that end in caller (pageable) memory: orbx::fetch_sync
    size_t area_pool = 0, area_last_total = 0;
    orbx::DevBuf<int> d_si[8];
    orbx::DevBuf<long long> d_qtprof;
    // k_quadtree_spill: node pool of the levels whose quadtree does not fit the LDS (orbx_api.cpp: quadtree_plan); qt_lds_nodes = largest tree the LDS form
    // is given (orbx_debug_quadtree_lds_nodes: tests lower it to run small cases through the pool form)
    orbx::DevBuf<uint8_t> d_qtpool; int qt_lds_nodes = orbx::kQuadtreeLdsNodes, cfg_qt_lds_nodes = 0, qt_pool_levels = 0;      // qt_pool_levels: levels the last extraction ran in the pool form
    // hipGraph replay of the extraction pipeline (orbx_set_graph_replay)
    bool use_graph = false;
#ifndef ORBX_EMU
    hipGraph_t graph = nullptr; hipGraphExec_t graph_exec = nullptr;
#endif
    int g_B = 0, g_stride = 0, g_lap0 = 0, g_lap1 = 0, g_W = 0, g_H = 0, g_gauss = 0; size_t g_image_stride = 0;
    const void* g_images = nullptr; const void* g_pyr = nullptr;
    // input pre-step (orbx_set_input): channels / colour order / grey coefficients / geometry, device maps or taps, intermediate frame
    bool in_active = false; int in_channels = 1, in_rgb = 1, in_gray_variant = 0, in_geometry = 0, in_out_w = 0, in_out_h = 0, in_tap_w = 0, in_tap_h = 0;
    orbx::DevBuf<float> d_mapx, d_mapy; orbx::DevBuf<orbx::ResizeTap> d_in_xt, d_in_yt; orbx::DevBuf<uint8_t> d_frame;
    orbx::DevBuf<int> d_rowstart, d_rowitems;   // row index of every image's keypoints for the stereo search (k_layout -> k_stereo_match)
    // batched SearchLocalPoints (orbm_search_local_points_batch): one device block, one pinned result block, the size of the last enqueue
    orbx::DevBuf<uint8_t> d_lp, d_depth_in; orbx::HostBuf<uint8_t> h_lp_in, h_lp_out;
    size_t lp_pool = 0; int lp_B = 0, lp_M = 0, lp_first = 0; size_t lp_o_counter = 0, lp_o_view = 0; bool lp_pending = false, lp_want_view = false;
    orbx::rt::event_t ev_lp = 0;
    // Frame::UndistortKeyPoints on the device (orbx_set_undistort): mvKeysUn of the last batch
    orbx::UndistortParams undist = {}; int undist_gen = 0, g_undist_gen = 0; orbx::DevBuf<orbx::KeyPointRec> d_kps_un;
    // the model the LAST EXTRACTION ran with: d_kps_un holds that batch's mvKeysUn (or nothing when it ran without a model).  Consumers read these,
    // not the live `undist` - orbx_set_undistort between an extraction and a search would otherwise point them at stale or unallocated keypoints
    int ex_undist_gen = 0; bool ex_undist_active = false;
    orbx::DevBuf<int> d_aux;     // int4 per keypoint: stereo row band / x / octave (k_orient_brief -> k_stereo_match)
};
// mvKeysUn of the last extraction no longer matches the undistortion model of the handle (orbx_set_undistort was called in between)
inline void record_done_if_pending(orbx_extractor* h) { if (h->done_lazy) { orbx::rt::event_record(h->ev_done, h->s0); h->done_lazy = false; } }
inline void record_import_if_pending(orbx_extractor* h) { if (h->import_lazy) { orbx::rt::event_record(h->ev_import, h->s0); h->import_lazy = false; } }
// Device results -> caller memory, waiting for them: the copy lands in page-locked memory of the handle and is moved on by the host.  A hipMemcpyAsync into
// pageable memory is staged by the runtime chunk by chunk behind a blocking wait of its own (a synchronous call of a few KB cost 33 us that way, 17 us this way:
// profiles/r05/sync_latency_probe.txt, next_rows.json).
inline int fetch_sync(orbx_extractor* h, void* dst, const void* dev_src, size_t bytes) {
    if (bytes == 0) return orbx::rt::stream_sync(h->s0);
    if (orbx::rt::memory_is_host()) { if (orbx::rt::stream_sync(h->s0)) return -1; memcpy(dst, dev_src, bytes); return 0; }
    if (h->h_res.ensure(bytes + 64)) return -1;
    if (orbx::rt::copy_d2h(h->h_res.p, dev_src, bytes, h->s0) || orbx::rt::stream_sync(h->s0)) return -1;
    memcpy(dst, h->h_res.p, bytes);
    return 0;
}
inline bool undistort_stale(const orbx_extractor* h) { return h->lastB > 0 && h->ex_undist_gen != h->undist_gen; }
