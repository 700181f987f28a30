// orbx_kernels.h — prototypes of the __global__ kernels (defined in k_*.hip) for the host launcher.
#pragma once
#include "orbx_types.h"
#include "undistort_model.h"

namespace orbx {

__global__ void k_import(const LevelInfo* __restrict__ lv, const uint8_t* __restrict__ images, int stride,
                         size_t image_stride, uint8_t* __restrict__ pyr, size_t pyr_stride);
__global__ void k_resize(const LevelInfo* __restrict__ lv, int level, const ResizeTap* __restrict__ xtab,
                         const ResizeTap* __restrict__ ytab, uint8_t* __restrict__ pyr, size_t pyr_stride,
                         int lds_pitch, int lds_rows);
// strip_rows (<= 64): output rows per wave; a block covers 256 columns x 4 strips
__global__ void k_resize_rows(const LevelInfo* __restrict__ lv, int level, const ResizeTap* __restrict__ xtab,
                              const ResizeTap* __restrict__ ytab, uint8_t* __restrict__ pyr, size_t pyr_stride, int strip_rows);
#ifndef ORBX_PYR_THREADS
#define ORBX_PYR_THREADS 512
#endif
#ifndef ORBX_PYR_TILE
#define ORBX_PYR_TILE 16
#endif
constexpr int kPyrTile = ORBX_PYR_TILE;        // k_pyramid_fused: a workgroup's tile of the top pyramid level
constexpr int kPyrThreads = ORBX_PYR_THREADS;  // ... and its threads (the tile is a latency problem: many waves per tile)
__global__ void k_pyramid_fused(const LevelInfo* __restrict__ lv, int nlevels, const ResizeTap* __restrict__ xtab, const ResizeTap* __restrict__ ytab,
                                const PyrSpan* __restrict__ xspan, const PyrSpan* __restrict__ yspan, int ntx, uint8_t* __restrict__ pyr, size_t pyr_stride,
                                int buf_a_bytes, int buf_b_bytes, PyrTapOffsets toff);
#ifndef ORBX_FAST_XCD_RUN
#define ORBX_FAST_XCD_RUN 4
#endif
constexpr int kFastXcdRun = ORBX_FAST_XCD_RUN;   // neighbouring FAST cells kept on one XCD (k_fast_cells)
constexpr int kFastThreadsDecl = 64;   // must equal kFastThreads in k_fast.hip
constexpr int kFastPitch = 48;         // LDS pitch of the FAST window tile for cells whose dword-aligned window fits in it (cells up to 39 px wide)
__global__ void k_fast_cells(const LevelInfo* __restrict__ lv, const CellInfo* __restrict__ cells, int ncells,
                             const uint8_t* __restrict__ pyr, size_t pyr_stride, int iniTh, int minTh,
                             uint32_t* __restrict__ slots, size_t slots_stride, int* __restrict__ cell_count,
                             int tile_bytes, int list_bytes, int* __restrict__ status);
constexpr int kResizeRows = 8;         // output rows per k_resize tile (256 columns wide)
// Output rows per k_blur wave (a block covers 256 columns x 4 strips of that many rows).  A strip reads 6 halo rows on top of its own: large batches
// run strips of 32 rows (the horizontal pass of 38 input rows per 32 outputs instead of 22 per 16: +1.0 % on the headline, profiles/r06/ab_experiments.txt),
// small batches - where the number of waves in flight is what counts, and the blur shares its launch with the FAST cells - strips of 16.
#ifndef ORBX_BLUR_ROWS
#define ORBX_BLUR_ROWS 16
#endif
#ifndef ORBX_BLUR_ROWS_LARGE
#define ORBX_BLUR_ROWS_LARGE 32
#endif
constexpr int kBlurRows = ORBX_BLUR_ROWS, kBlurRowsLarge = ORBX_BLUR_ROWS_LARGE;
constexpr int kSimdSelftestOps = 22;
__global__ void k_simd_selftest(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ c, int n, uint32_t* __restrict__ out);
__global__ void k_blur(const LevelInfo* __restrict__ lv, int nlevels, const uint8_t* __restrict__ pyr,
                       uint8_t* __restrict__ blur, size_t pyr_stride, BlurTaps taps, BlurTiles tiles);
__global__ void k_blur_large(const LevelInfo* __restrict__ lv, int nlevels, const uint8_t* __restrict__ pyr,
                             uint8_t* __restrict__ blur, size_t pyr_stride, BlurTaps taps, BlurTiles tiles);     // strips of kBlurRowsLarge rows
constexpr int kQuadtreeThreads = 1024; // workgroup size of k_quadtree; a level uses its first LevelInfo::qt_threads threads
// small batches: the blur strips and the FAST cells of an image in one launch (k_fast.hip)
__global__ void k_fast_cells_blur(const LevelInfo* __restrict__ lv, const CellInfo* __restrict__ cells, int ncells,
                                  const uint8_t* __restrict__ pyr, size_t pyr_stride, int iniTh, int minTh,
                                  uint32_t* __restrict__ slots, size_t slots_stride, int* __restrict__ cell_count,
                                  int tile_bytes, int list_bytes, int* __restrict__ status,
                                  int nlevels, uint8_t* __restrict__ blur, BlurTaps taps, BlurTiles tiles, int blur_waves);
__global__ void k_quadtree(const LevelInfo* __restrict__ lv, const CellInfo* __restrict__ cells, int ncells,
                           const int* __restrict__ cell_count, const uint32_t* __restrict__ slots, size_t slots_stride,
                           uint32_t* __restrict__ candA, uint32_t* __restrict__ candB, size_t cand_stride,
                           uint32_t* __restrict__ lvl_keys, int kp_total_cap, int* __restrict__ lvl_count,
                           int nlevels, int node_cap, int nb_cap, int lut_x, int lut_y, int* __restrict__ status, long long* __restrict__ qt_prof,
                           int wide, int counter_bytes, int level0);
// levels whose node lists exceed the LDS (the 5 x nFeatures extractor of the monocular initialisation): node arrays in a global pool
__global__ void k_quadtree_spill(const LevelInfo* __restrict__ lv, const CellInfo* __restrict__ cells, int ncells,
                           const int* __restrict__ cell_count, const uint32_t* __restrict__ slots, size_t slots_stride,
                           uint32_t* __restrict__ candA, uint32_t* __restrict__ candB, size_t cand_stride,
                           uint32_t* __restrict__ lvl_keys, int kp_total_cap, int* __restrict__ lvl_count,
                           int nlevels, int node_cap, int nb_cap, int lut_x, int lut_y, int* __restrict__ status, long long* __restrict__ qt_prof,
                           int wide, int counter_bytes, unsigned char* __restrict__ pool, size_t pool_stride);
constexpr int kQuadtreeLdsNodes = 4000;     // the LDS form packs positions of its sort ranges in 12 bits: trees with more nodes take the pool form whatever the LDS
__global__ void k_layout(const LevelInfo* __restrict__ lv, int nlevels, const uint32_t* __restrict__ lvl_keys,
                         int kp_total_cap, const int* __restrict__ lvl_count, int lap0, int lap1,
                         int* __restrict__ final_idx, int* __restrict__ n_out, int* __restrict__ mono_out,
                         int nb, int* __restrict__ row_start, int* __restrict__ row_items);
#ifndef ORBX_STEREO_ROW_SHIFT
#define ORBX_STEREO_ROW_SHIFT 3
#endif
constexpr int kStereoRowShift = ORBX_STEREO_ROW_SHIFT;   // k_layout / k_stereo_match: right keypoints are bucketed by (first row of their band) >> shift
constexpr int kKpPerWaveDecl = 8;      // must equal kKpPerWave in k_describe.hip
constexpr int kKpPerWaveSmallDecl = 2; // keypoints per wave of k_orient_brief_small
__global__ void k_orient_brief(const LevelInfo* __restrict__ lv, int nlevels, const uint8_t* __restrict__ pyr,
                               const uint8_t* __restrict__ blur, size_t pyr_stride,
                               const uint32_t* __restrict__ lvl_keys, int kp_total_cap,
                               const int* __restrict__ lvl_count, const int* __restrict__ final_idx, UmaxTab umax,
                               KeyPointRec* __restrict__ out_kps, unsigned long long* __restrict__ out_desc, int4* __restrict__ out_aux,
                               int B, int groups_per_image);
__global__ void k_orient_brief_small(const LevelInfo* __restrict__ lv, int nlevels, const uint8_t* __restrict__ pyr,
                               const uint8_t* __restrict__ blur, size_t pyr_stride,
                               const uint32_t* __restrict__ lvl_keys, int kp_total_cap,
                               const int* __restrict__ lvl_count, const int* __restrict__ final_idx, UmaxTab umax,
                               KeyPointRec* __restrict__ out_kps, unsigned long long* __restrict__ out_desc, int4* __restrict__ out_aux,
                               int B, int groups_per_image);
__global__ void k_undistort(const KeyPointRec* __restrict__ kps, const int* __restrict__ n_per_frame, int cap, UndistortParams U, KeyPointRec* __restrict__ kps_un);
__global__ void k_hamming_matrix(const unsigned long long* __restrict__ A, int na,
                                 const unsigned long long* __restrict__ Bm, int nb, int* __restrict__ out);
__global__ void k_stereo_match(const LevelInfo* __restrict__ lv, const KeyPointRec* __restrict__ kpsL,
                               const unsigned long long* __restrict__ descL, const int* __restrict__ nL,
                               const KeyPointRec* __restrict__ kpsR, const unsigned long long* __restrict__ descR,
                               const int4* __restrict__ auxR, const int* __restrict__ nR, const int* __restrict__ bucket_start,
                               const int* __restrict__ bucket_items, int nb, int lookback, int cap, const uint8_t* __restrict__ pyrL,
                               const uint8_t* __restrict__ pyrR, size_t pyr_stride, StereoParams P,
                               float* __restrict__ uRight, float* __restrict__ depth, int* __restrict__ sad);
__global__ void k_stereo_median(const int* __restrict__ nL, int cap, float* __restrict__ uRight,
                                float* __restrict__ depth, const int* __restrict__ sad, int* __restrict__ n_matches);
__global__ void k_kb8_stereo(const KeyPointRec* __restrict__ kpsL, const int* __restrict__ monoL, const int* __restrict__ nL,
                             const KeyPointRec* __restrict__ kpsR, const int* __restrict__ monoR, int cap, const int* __restrict__ idx0,
                             const uint8_t* __restrict__ ratio_ok, KB8StereoParams P, int* __restrict__ l2r, int* __restrict__ r2l,
                             float* __restrict__ depth, float* __restrict__ p3d, int* __restrict__ nmatches);
__global__ void k_knn2_mfma(const unsigned long long* __restrict__ descQ, const int* __restrict__ qoff, const int* __restrict__ nq,
                            const unsigned long long* __restrict__ descT, const int* __restrict__ toff, const int* __restrict__ nt, int cap,
                            int* __restrict__ idx0, int* __restrict__ dist0, int* __restrict__ idx1, int* __restrict__ dist1, uint8_t* __restrict__ ratio_ok);
__global__ void k_knn2(const unsigned long long* __restrict__ descQ, const int* __restrict__ qoff, const int* __restrict__ nq,
                       const unsigned long long* __restrict__ descT, const int* __restrict__ toff, const int* __restrict__ nt,
                       int cap, int* __restrict__ idx0, int* __restrict__ dist0, int* __restrict__ idx1,
                       int* __restrict__ dist1, uint8_t* __restrict__ ratio_ok);

constexpr int kGridThreads = 1024;     // workgroup size of k_grid_build (3 grid cells per thread)
constexpr int kGridCellStride = 64 * 48 + 2;   // ints per frame of the batched cell_start arrays
__global__ void k_grid_build(const KeyPointRec* __restrict__ kps, int N, GridParams g, int* __restrict__ cell_of,
                             int* __restrict__ cell_start, int* __restrict__ cell_items, const int* __restrict__ n_per_frame, int frame_stride);
constexpr int kAreaWaves = 16;         // queries (waves) per k_area_search workgroup
__global__ void k_area_search(const AreaQuery* __restrict__ queries, const unsigned long long* __restrict__ qdesc, int Q,
                              const KeyPointRec* __restrict__ kps, const float* __restrict__ u_right,
                              const unsigned long long* __restrict__ fdesc, GridParams g, const int* __restrict__ cell_start,
                              const int* __restrict__ cell_items, int gate_right, int* __restrict__ pool_counter, int pool_cap,
                              int* __restrict__ q_start, int* __restrict__ q_count, int2* __restrict__ entries, int frame_stride);
__global__ void k_frustum(FrustumParams F, int M, const float* __restrict__ pos, const float* __restrict__ normal, const float* __restrict__ min_dist,
                          const float* __restrict__ max_dist, const uint8_t* __restrict__ is_bad, uint8_t* __restrict__ in_view, float* __restrict__ track,
                          int* __restrict__ scale_level, AreaQuery* __restrict__ queries, int* __restrict__ zero4, const FrustumParams* __restrict__ Fbatch);
__global__ void k_project_points(ProjectParams P, int M, const float* __restrict__ pos, const float* __restrict__ normal, const float* __restrict__ min_inv,
                                 const float* __restrict__ max_inv, const uint8_t* __restrict__ skip, uint8_t* __restrict__ valid, float* __restrict__ out, int debug_flags);
__global__ void k_area_search_threads(const AreaQuery* __restrict__ queries, const unsigned long long* __restrict__ qdesc, int Q,
                                      const KeyPointRec* __restrict__ kps, const float* __restrict__ u_right,
                                      const unsigned long long* __restrict__ fdesc, GridParams g, const int* __restrict__ cell_start,
                                      const int* __restrict__ cell_items, int gate_right, int* __restrict__ pool_counter, int pool_cap,
                                      int* __restrict__ q_start, int* __restrict__ q_count, int2* __restrict__ entries, int frame_stride, int qdesc_per_frame);
__global__ void k_lastframe_queries(const FrustumParams* __restrict__ Fb, int capL, const int* __restrict__ n_last, const float* __restrict__ pos,
                                    const uint8_t* __restrict__ valid, const int* __restrict__ octave, AreaQuery* __restrict__ queries, int* __restrict__ zero4);
__global__ void k_keyframe_queries(const FrustumParams* __restrict__ Fb, int capL, const int* __restrict__ n_kf, const float* __restrict__ pos,
                                   const uint8_t* __restrict__ valid, const float* __restrict__ min_dist, const float* __restrict__ max_dist,
                                   AreaQuery* __restrict__ queries, int* __restrict__ zero4);
__global__ void k_lastframe_accept(int M, int cap, const int* __restrict__ n_per_frame, const int* __restrict__ q_start, const int* __restrict__ q_count,
                                   const int2* __restrict__ entries, const uint8_t* __restrict__ occupied0, const uint8_t* __restrict__ has_obs, int th_high,
                                   int* __restrict__ assigned, int* __restrict__ nmatches, const float* __restrict__ last_angle,
                                   const KeyPointRec* __restrict__ cur_kps, int check_ori);
__global__ void k_local_accept(int M, int cap, const int* __restrict__ n_per_frame, const int* __restrict__ q_start, const int* __restrict__ q_count,
                               const int2* __restrict__ entries, const uint8_t* __restrict__ occupied0, const uint8_t* __restrict__ has_obs, float nnratio,
                               int th_high, int* __restrict__ assigned, int* __restrict__ nmatches);
__global__ void k_stereo_from_depth(const KeyPointRec* __restrict__ kps, const KeyPointRec* __restrict__ kps_un, const int* __restrict__ n_per_frame, int cap,
                                    const float* __restrict__ depth, int stride, size_t image_stride, int w, int h, float mbf, float* __restrict__ u_right,
                                    float* __restrict__ depth_out, int* __restrict__ n_valid);
__global__ void k_bow_search(const BowItem* __restrict__ items, int nitems, const KeyPointRec* __restrict__ kps1,
                             const unsigned long long* __restrict__ desc1, const float* __restrict__ ur1,
                             const KeyPointRec* __restrict__ kps2, const unsigned long long* __restrict__ desc2,
                             const float* __restrict__ ur2, const uint8_t* __restrict__ has_mp2, const int* __restrict__ feat2,
                             const BowParams* __restrict__ Ps, int* __restrict__ best2);
__global__ void k_bow_search_kb8(const BowItem* __restrict__ items, int nitems, const KeyPointRec* __restrict__ kps1,
                             const unsigned long long* __restrict__ desc1, const float* __restrict__ ur1,
                             const KeyPointRec* __restrict__ kps2, const unsigned long long* __restrict__ desc2,
                             const float* __restrict__ ur2, const uint8_t* __restrict__ has_mp2, const int* __restrict__ feat2,
                             const BowParams* __restrict__ Ps, int* __restrict__ best2);
__global__ void k_sft_resident(ResidentKF k1, const uint8_t* __restrict__ flags, const SftNeighbour* __restrict__ nb, int* __restrict__ best);
__global__ void k_sft_resident_kb8(ResidentKF k1, const uint8_t* __restrict__ flags, const SftNeighbour* __restrict__ nb, int* __restrict__ best);
__global__ void k_bow_match_resident(const BowPairResident* __restrict__ pairs, const uint8_t* __restrict__ flags, float nnratio, int th_low,
                                     int th_inclusive, int* __restrict__ m12, int N1cap, int* __restrict__ status);
__global__ void k_bow_rotation_prune(const BowPairResident* __restrict__ pairs, int* __restrict__ m12, int N1cap, int check_ori, int* __restrict__ nmatches);
__global__ void k_bow_dists(const BowItem* __restrict__ items, int nitems, const unsigned long long* __restrict__ desc1,
                            const unsigned long long* __restrict__ desc2, const uint8_t* __restrict__ eligible2,
                            const int* __restrict__ feat2, int* __restrict__ out);

__global__ void k_distinctive(const unsigned long long* __restrict__ desc, const int* __restrict__ start, int P, int* __restrict__ best);
__global__ void k_voc_descend(const unsigned long long* __restrict__ fdesc, const int* __restrict__ n_feat, int n_fixed, int cap, int B,
                              const unsigned long long* __restrict__ slot_desc, const VocSlot* __restrict__ slots,
                              const double* __restrict__ slot_weight, int root_children, int nid_level, unsigned* __restrict__ out_word,
                              unsigned* __restrict__ out_node, double* __restrict__ out_weight);
__global__ void k_voc_assemble(const unsigned* __restrict__ word, const unsigned* __restrict__ node, const double* __restrict__ weight,
                               const int* __restrict__ n_feat, int n_fixed, int cap, int P, int weighting, int norm,
                               unsigned* __restrict__ bow_id, double* __restrict__ bow_val, int* __restrict__ bow_start,
                               unsigned* __restrict__ fv_node, int* __restrict__ fv_start, unsigned* __restrict__ fv_feat,
                               int* __restrict__ n_out);

__global__ void k_input_remap(const uint8_t* __restrict__ src, int sw, int sh, int sstride, size_t simg, int C, const float* __restrict__ mapx,
                              const float* __restrict__ mapy, int out_w, int out_h, uint8_t* __restrict__ dst, int dst_pitch, size_t dst_stride);
__global__ void k_input_resize(const uint8_t* __restrict__ src, int sw, int sh, int sstride, size_t simg, int C, const ResizeTap* __restrict__ xt,
                               const ResizeTap* __restrict__ yt, int out_w, int out_h, uint8_t* __restrict__ dst, int dst_pitch, size_t dst_stride);
__global__ void k_input_gray(const uint8_t* __restrict__ src, int sstride, size_t simg, int C, int ridx, int ry, int gy, int by, int shift, int w, int h,
                             uint8_t* __restrict__ dst, int dst_pitch, size_t dst_stride);

}  // namespace orbx
