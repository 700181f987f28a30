// orbx_types.h — device-visible tables and packed record formats shared by the kernels and the host.
#pragma once
#include <cstdint>
#include "orbx_platform.h"

namespace orbx {

constexpr int kMaxLevels = 16;
constexpr int kEdge = 19;        // EDGE_THRESHOLD, reference src/ORBextractor.cc:78
constexpr int kBorder = 16;      // minBorderX/Y = EDGE_THRESHOLD-3, :1076-1077
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE, :77

// Candidate / keypoint key: x (12 bits) | y (12 bits) << 12 | score (8 bits) << 24.
// FAST candidates carry coordinates relative to the detection border (x-16, y-16) exactly like the
// reference's vToDistributeKeys (:1159-1160); quadtree outputs keep the same packing.
__host__ __device__ inline uint32_t key_pack(int x, int y, int s) { return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)s << 24); }
__host__ __device__ inline int key_x(uint32_t k) { return (int)(k & 0xFFFu); }
__host__ __device__ inline int key_y(uint32_t k) { return (int)((k >> 12) & 0xFFFu); }
__host__ __device__ inline int key_s(uint32_t k) { return (int)(k >> 24); }

struct LevelInfo {
    int w, h, pitch;         // level image size and row pitch (bytes, multiple of 64)
    int off;                 // byte offset of this level inside one image's pyramid block
    float scale, inv_scale;  // mvScaleFactor / mvInvScaleFactor (:478-500)
    int quota;               // mnFeaturesPerLevel (:509-526)
    int patch;               // scaledPatchSize = int(31*scale) (:1184)
    int kp_cap, kp_off;      // capacity / offset of this level in the per-image level-ordered keypoint arrays
    int ncols, nrows, wcell, hcell;   // FAST cell grid (:1087-1095)
    int cell_begin, cell_count;       // this level's cells in the flattened cell table
    int cand_cap, cand_off;  // capacity / offset of this level in the per-image candidate arrays
    int bw, bh;              // detection-area size (maxBorder-minBorder), the quadtree's root extent
    int nini;                // number of quadtree roots = round(bw/bh) (:718)
    float hX;                // root width = bw/nini (:722)
    int xtab_off, ytab_off;  // offsets into the resize coefficient tables (levels >= 1)
    int presort_depth;       // quadtree: keys are counting-sorted by (root, first presort_depth child digits) up front
    int qt_threads;          // quadtree: threads of the workgroup that work on this level (256, or 1024 for the large levels)
};

struct CellInfo {
    int16_t level;
    int16_t x0, y0, x1, y1;  // detectable interior of the cell window in level coordinates
    int16_t _pad;
    int slot_off;            // first candidate slot of this cell inside the per-image slot array
};

// Resize coefficient entry: source index + two 11-bit fixed-point weights (a0 | a1 << 16).
struct ResizeTap { int ofs; int w; };

// k_pyramid_fused: what one tile column (or row) of the top level computes and owns at one level: region [a, b) (for columns a and b are
// multiples of 4), owned interval [o0, o1).  Index: tile * nlevels + level; level 0 carries the window to load (nothing of it is written).
struct PyrSpan { int16_t a, b, o0, o1; };
struct PyrTapOffsets { int x[kMaxLevels], y[kMaxLevels], total; };   // first LDS tap entry of a level's region columns / rows (sized for the widest tile)

struct KeyPointRec { float x, y, size, angle, response; int32_t octave, class_id; };   // = OrbxKeyPoint, 28 B

struct BlurTaps { int k[7]; };                       // 7-tap Gaussian, 8-bit fixed point
struct BlurTiles { int begin[kMaxLevels + 1]; };         // first tile (256 cols x 64 rows) of each level in k_blur's grid
struct UmaxTab { int u[16]; };                        // circular patch half-widths (src/ORBextractor.cc:542-570)
struct StereoParams { float mbf, mb; int th_high, th_orb; int debug_flags; };   // ORBmatcher::TH_HIGH, (TH_HIGH+TH_LOW)/2

// ---- guided searches (k_search.hip) ----
struct GridParams {                                               // mnMinX, mnMinY, mfGridElementWidthInv/HeightInv (include/Frame.h:250-251)
    float min_x, min_y, gw_inv, gh_inv;
    float inv_sigma2[kMaxLevels];                                 // mvInvLevelSigma2, read by the chi-square gate (AreaQuery::gate == 2) only
};
struct AreaQuery {                                                // one GetFeaturesInArea call + the per-candidate gate
    float x, y, r, ur;
    int min_level, max_level, active, gate;                      // gate: 0 none, 1 right coordinate (ORBmatcher.cc:107-117), 2 Fuse chi-square (:1437-1469)
};
struct FrustumParams {                                            // what Frame::isInFrustum reads of the Frame (src/Frame.cc:667-773)
    float Rcw[9], tcw[3], Ow[3];                                  // mRcw (row-major), mtcw, mOw
    float qcw[4];                                                 // mTcw.unit_quaternion().coeffs() (x, y, z, w): `Tcw * x3Dw` of the LastFrame search (sophus_action.h)
    float cam[8]; int kb8;                                        // mpCamera: pinhole fx, fy, cx, cy or the 8 Kannala-Brandt parameters
    float min_x, max_x, min_y, max_y, mbf;
    float log_scale_factor; int nlevels;                          // mfLogScaleFactor, mnScaleLevels (MapPoint::PredictScale)
    float scale_factors[kMaxLevels];
    float cos_limit;                                              // viewingCosLimit
    // SearchByProjection(Frame, MapPoints) query parameters when the queries are produced on the device
    float th, th_far; int far_points;
    int rig_mode;                                                 // Frame::isInFrustumChecks: store nothing unless every test passes, level -1 when rejected
    int forward, backward;                                        // SearchByProjection(Frame, LastFrame): bForward / bBackward (k_lastframe_queries)
    int debug_flags;                                              // orbx_debug_stereo_flags (tests): bit 4 = matrix form of Tcw * p (sophus_action.h)
};
// orbm_project_points: the geometry in front of GetFeaturesInArea in the projection-type searches (ORBmatcher.cc:495-732, :1325-1675, :1689-1932,
// :1950-2030, :2196-2260) - see OrbmProjection in include/orbx.h (same fields)
struct ProjectParams {
    float q[4], t[3];
    int second; float q2[4], t2[3], s2;
    float Ow[3];
    int dist_mode, depth_test, camera_type; float cam[8]; int inline_pinhole;
    float min_x, max_x, min_y, max_y; int bounds_mode;
    int distance_test, angle_test;
    float bf;
};
struct VocSlot { int node_id, child_start, child_cnt, word_id; };   // one vocabulary node; children occupy consecutive slots
struct BowItem { int idx1, start2, cnt2, out_off; };
struct BowParams {
    float F12[9];            // fundamental matrix, row-major (Pinhole::epipolarConstrain)
    float ep[2];             // epipole of KF1's centre in KF2
    float scale2[kMaxLevels], sigma2_2[kMaxLevels];
    int only_stereo, coarse, th_low;
    // Kannala-Brandt cameras (KannalaBrandt8::epipolarConstrain = TriangulateMatches > 1e-4): kb8 != 0
    int kb8, nleft1, nleft2;                 // NLeft of the two key frames (-1: one camera)
    float sigma2_1[kMaxLevels];              // mvLevelSigma2 of KF1
    float cam1[2][8], cam2[2][8];            // mvParameters of mpCamera / mpCamera2 of KF1 and KF2
    float R[4][9], t[4][3];                  // relative pose by [bRight1 * 2 + bRight2] (Tll, Tlr, Trl, Trr); one camera: entry 0
};
// Device-resident arrays of one key frame / frame (orbm_keyframe): what the vocabulary-bucket searches read of it, uploaded once.
struct ResidentKF {
    const KeyPointRec* kps; const unsigned long long* desc; const float* ur;
    const uint32_t* node_id;     // mFeatVec as CSR: node ids (ascending), fv_nodes + 1 offsets, feature indices
    const int* fv_start; const int* fv_feat;
    const int* node_of_feat;     // feature -> index of its node in node_id (-1: in no node)
    int N, fv_nodes;
};
struct SftNeighbour { ResidentKF k2; BowParams P; int mp2_off; int _pad; };        // one neighbour of orbm_search_for_triangulation_resident
// one pair of orbm_search_by_bow_resident (offsets into the flag block, -1 = none / all).  k2_nodes_dev != NULL: K2 is a FRAME of the last extraction
// whose FeatureVector the vocabulary transform left on the device; its node count is read there (orbm_search_by_bow_frames_batch)
struct BowPairResident { ResidentKF k1, k2; int mp1_off, elig2_off; const int* k2_nodes_dev; };
struct KB8StereoParams {                     // Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1530-1587)
    float cam1[8], cam2[8];                  // mpCamera, mpCamera2
    float R12[9], t12[3];                    // mRlr, mtlr
    float sigma2[kMaxLevels];                // mvLevelSigma2
};
#ifdef ORBX_EMU
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
#endif

// quadtree node, 24 B, lives in LDS
struct QNode {
    int16_t x0, y0, x1, y1;
    uint32_t start;     // first key of this node's span in the key buffers
    uint32_t cnt_buf;   // bits 0..29 key count, bit 30 = which key buffer (0:A 1:B) holds the span
    uint32_t code;      // path from the root: root index, then one base-4 digit (child n1..n4) per level
    uint32_t depth;     // number of digits; while depth < the level's presort depth the span is already ordered by the next digit
};

}  // namespace orbx
