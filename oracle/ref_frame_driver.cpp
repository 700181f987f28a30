// oracle/ref_frame_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// C-ABI wrapper around the REFERENCE's own ORB_SLAM3::Frame, compiled from /root/reference/src/Frame.cc + src/ORBextractor.cc (unmodified,
// read in place; never copied into this repo) against oracle/opencv_shim and oracle/slam_shim/frame_world.h.  Built by oracle/Makefile
// into oracle/_ref/libref_frame.so.  One entry point runs the reference's stereo Frame constructor (src/Frame.cc:105-230): two
// ORBextractors on two threads, UndistortKeyPoints (no distortion), ComputeStereoMatches (:1102-1358), AssignFeaturesToGrid (:469-503);
// the others read the resulting Frame and call its GetFeaturesInArea (:859-951).
#include <cstdint>
#include <chrono>
#include <cstring>
#include <new>
#include <vector>
#include "Frame.h"          // the reference header, via -I/root/reference/include
#include "ORBextractor.h"
#include "ORBmatcher.h"

using namespace ORB_SLAM3;

namespace {
struct RefKp { float x, y, size, angle, response; int octave, class_id; };
struct Holder {
    ORBextractor *left = nullptr, *right = nullptr; Pinhole* cam = nullptr; KannalaBrandt8 *kb1 = nullptr, *kb2 = nullptr; Frame* frame = nullptr;
    ~Holder() { delete frame; delete left; delete right; delete cam; delete kb1; delete kb2; }
};
// a pose handed over as (R row-major, t) enters through Sophus' SE3(R, t) constructor (matrix -> unit quaternion, se3.hpp:480-482)
Sophus::SE3f se3_from(const float* R, const float* t) {
    Eigen::Matrix3f Rm; Eigen::Vector3f tv;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rm(i, j) = R[3 * i + j]; tv(i) = t[i]; }
    return Sophus::SE3f(Rm, tv);
}
void put_keys(const std::vector<cv::KeyPoint>& k, void* out) {
    RefKp* o = (RefKp*)out;
    for (size_t i = 0; i < k.size(); i++) o[i] = {k[i].pt.x, k[i].pt.y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id};
}
}  // namespace

extern "C" {

// Frame(imLeft, imRight, ...) with fresh extractors and a pinhole camera without distortion.  Returns a handle; *n / *n_right = keypoint counts.
void* ref_frame_stereo(const uint8_t* L, const uint8_t* R, int w, int h, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int gauss_variant,
                       float fx, float fy, float cx, float cy, float bf, float th_depth, int* n, int* n_right) {
    cv::shim_gauss_variant() = gauss_variant;
    Holder* H = new Holder();
    H->left = new ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th);
    H->right = new ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th);
    H->cam = new Pinhole(std::vector<float>{fx, fy, cx, cy});
    cv::Mat imL(h, w, CV_8UC1, (void*)L, (size_t)w), imR(h, w, CV_8UC1, (void*)R, (size_t)w);
    cv::Mat K = H->cam->toK();
    cv::Mat dist(4, 1, CV_32F); for (int i = 0; i < 4; i++) dist.at<float>(i) = 0.0f;
    Frame::mbInitialComputations = true;                 // image bounds and grid constants are per image size (:185-203)
    // The reference's constructor calls ComputeStereoMatches() (src/Frame.cc:164), which reads the member mb (:1163 minZ = mb), BEFORE it
    // assigns mb = mbf / fx (:206); mb is not in the initialiser list.  In the running system every Frame is a temporary built in the same
    // stack slot of Tracking::GrabImageStereo (src/Tracking.cc:1565-1582), so from the second frame on the value read is the one the previous
    // frame left there, mbf / fx.  The driver reproduces that state: raw storage whose mb field already holds mbf / fx.
    void* storage = ::operator new(sizeof(Frame));
    memset(storage, 0, sizeof(Frame));
    reinterpret_cast<Frame*>(storage)->mb = bf / fx;
    H->frame = new (storage) Frame(imL, imR, 0.0, H->left, H->right, nullptr, K, dist, bf, th_depth, H->cam);
    *n = H->frame->N; *n_right = (int)H->frame->mvKeysRight.size();
    return H;
}
void ref_frame_destroy(void* h) { delete (Holder*)h; }

// Frame(imGray, imDepth, ...) - the RGB-D constructor (src/Frame.cc:235-345): one extraction, UndistortKeyPoints (no distortion),
// ComputeStereoFromRGBD (:1361-1391), AssignFeaturesToGrid.  depth: CV_32F, w x h (already scaled by the depth map factor).
void* ref_frame_rgbd(const uint8_t* gray, const float* depth, int w, int h, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int gauss_variant,
                     float fx, float fy, float cx, float cy, float bf, float th_depth, const float* dist5, int* n) {
    cv::shim_gauss_variant() = gauss_variant;
    Holder* H = new Holder();
    H->left = new ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th);
    H->cam = new Pinhole(std::vector<float>{fx, fy, cx, cy});
    cv::Mat im(h, w, CV_8UC1, (void*)gray, (size_t)w), imD(h, w, CV_32F, (void*)depth, (size_t)w * sizeof(float));
    cv::Mat K = H->cam->toK();
    // dist5 = (k1, k2, p1, p2, k3) as Examples/RGB-D/TUM1.yaml gives them: UndistortKeyPoints (:1003-1034) and ComputeImageBounds (:1043-1075) then
    // go through cv::undistortPoints (restated in the shim); NULL = no distortion
    cv::Mat dist(dist5 ? 5 : 4, 1, CV_32F); for (int i = 0; i < dist.rows; i++) dist.at<float>(i) = dist5 ? dist5[i] : 0.0f;
    Frame::mbInitialComputations = true;
    H->frame = new Frame(im, imD, 0.0, H->left, nullptr, K, dist, bf, th_depth, H->cam);
    *n = H->frame->N;
    return H;
}

// The fisheye-rig constructor (src/Frame.cc:1432-1528): two extractions with the cameras' lapping areas, ComputeStereoFishEyeMatches
// (:1530-1587: BFMatcher 2-NN on the lapping parts + the 0.7 ratio test + the triangulation gate),
// vconcat of the descriptors, AssignFeaturesToGrid.  out = {Nleft, Nright, monoLeft, monoRight}.
// cams = {cam1[8], cam2[8], Rlr[9] row-major, tlr[3]}: the gate is the reference's own KannalaBrandt8::TriangulateMatches
// (src/CameraModels/KannalaBrandt8.cpp:439-523, compiled unmodified into this library).  cams == NULL is served only by the ORBX_KB8_ACCEPT_ALL
// build of this driver (oracle/_ref/libref_frame_knn.so): there KannalaBrandt8.cpp is NOT linked and the class's members are defined below with a
// TriangulateMatches that accepts every pair, so that what the reference's loop leaves in mvLeftToRightMatch / mvRightToLeftMatch is exactly
// its kNN + ratio decision (TriangulateMatches is not virtual: it cannot be overridden in a subclass).
#ifdef ORBX_KB8_ACCEPT_ALL
namespace ORB_SLAM3 {
cv::Point2f KannalaBrandt8::project(const cv::Point3f&) { abort(); }
Eigen::Vector2d KannalaBrandt8::project(const Eigen::Vector3d&) { abort(); }
Eigen::Vector2f KannalaBrandt8::project(const Eigen::Vector3f&) { abort(); }
Eigen::Vector2f KannalaBrandt8::projectMat(const cv::Point3f&) { abort(); }
float KannalaBrandt8::uncertainty2(const Eigen::Matrix<double, 2, 1>&) { abort(); }
Eigen::Vector3f KannalaBrandt8::unprojectEig(const cv::Point2f&) { abort(); }
cv::Point3f KannalaBrandt8::unproject(const cv::Point2f&) { abort(); }
Eigen::Matrix<double, 2, 3> KannalaBrandt8::projectJac(const Eigen::Vector3d&) { abort(); }
bool KannalaBrandt8::ReconstructWithTwoViews(const std::vector<cv::KeyPoint>&, const std::vector<cv::KeyPoint>&, const std::vector<int>&, Sophus::SE3f&, std::vector<cv::Point3f>&,
                                             std::vector<bool>&) { abort(); }
cv::Mat KannalaBrandt8::toK() { abort(); }
Eigen::Matrix3f KannalaBrandt8::toK_() { abort(); }
bool KannalaBrandt8::epipolarConstrain(GeometricCamera*, const cv::KeyPoint&, const cv::KeyPoint&, const Eigen::Matrix3f&, const Eigen::Vector3f&, const float, const float) { abort(); }
bool KannalaBrandt8::matchAndtriangulate(const cv::KeyPoint&, const cv::KeyPoint&, GeometricCamera*, Sophus::SE3f&, Sophus::SE3f&, const float, const float, Eigen::Vector3f&) { abort(); }
float KannalaBrandt8::TriangulateMatches(GeometricCamera*, const cv::KeyPoint&, const cv::KeyPoint&, const Eigen::Matrix3f&, const Eigen::Vector3f&, const float, const float,
                                         Eigen::Vector3f& p3D) { p3D = Eigen::Vector3f(0, 0, 1); return 1.0f; }
}
#endif
void* ref_frame_fisheye(const uint8_t* L, const uint8_t* R, int w, int h, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int gauss_variant,
                        int lap_l0, int lap_l1, int lap_r0, int lap_r1, const float* cams, int* out) {
    cv::shim_gauss_variant() = gauss_variant;
    Holder* H = new Holder();
    H->left = new ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th);
    H->right = new ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th);
#ifdef ORBX_KB8_ACCEPT_ALL
    if (cams) { delete H; return nullptr; }
    H->kb1 = new KannalaBrandt8(std::vector<float>(8, 1.0f)); H->kb2 = new KannalaBrandt8(std::vector<float>(8, 1.0f));
#else
    if (!cams) { delete H; return nullptr; }
    H->kb1 = new KannalaBrandt8(std::vector<float>(cams, cams + 8)); H->kb2 = new KannalaBrandt8(std::vector<float>(cams + 8, cams + 16));
#endif
    H->kb1->mvLappingArea[0] = lap_l0; H->kb1->mvLappingArea[1] = lap_l1; H->kb2->mvLappingArea[0] = lap_r0; H->kb2->mvLappingArea[1] = lap_r1;
    cv::Mat imL(h, w, CV_8UC1, (void*)L, (size_t)w), imR(h, w, CV_8UC1, (void*)R, (size_t)w);
    cv::Mat K(3, 3, CV_32F); for (int i = 0; i < 9; i++) K.at<float>(i / 3, i % 3) = (i % 4 == 0) ? 1.0f : 0.0f;
    K.at<float>(0, 0) = 190.9f; K.at<float>(1, 1) = 190.9f; K.at<float>(0, 2) = 254.9f; K.at<float>(1, 2) = 256.9f;
    cv::Mat dist(4, 1, CV_32F); for (int i = 0; i < 4; i++) dist.at<float>(i) = 0.0f;
    Sophus::SE3f Tlr;
    if (cams) Tlr = se3_from(cams + 16, cams + 25);
    Frame::mbInitialComputations = true;
    H->frame = new Frame(imL, imR, 0.0, H->left, H->right, nullptr, K, dist, 19.3f, 40.0f, H->kb1, H->kb2, Tlr);
    out[0] = H->frame->Nleft; out[1] = H->frame->Nright; out[2] = H->frame->monoLeft; out[3] = H->frame->monoRight;
    return H;
}
// mvKeys [Nleft], mvKeysRight [Nright], mDescriptors [Nleft + Nright rows: left then right], mvLeftToRightMatch [Nleft], mvRightToLeftMatch [Nright]
void ref_frame_fisheye_get(void* h, void* keys, void* keys_right, uint8_t* desc, int* l2r, int* r2l) {
    Frame* F = ((Holder*)h)->frame;
    put_keys(F->mvKeys, keys); put_keys(F->mvKeysRight, keys_right);
    for (int i = 0; i < F->N; i++) memcpy(desc + 32 * (size_t)i, F->mDescriptors.ptr(i), 32);
    for (int i = 0; i < F->Nleft; i++) l2r[i] = F->mvLeftToRightMatch[i];
    for (int i = 0; i < F->Nright; i++) r2l[i] = F->mvRightToLeftMatch[i];
}
// mvDepth [Nleft], mvStereo3Dpoints [Nleft][3]
void ref_frame_fisheye_get3d(void* h, float* depth, float* p3d) {
    Frame* F = ((Holder*)h)->frame;
    for (int i = 0; i < F->Nleft; i++) { depth[i] = F->mvDepth[i]; for (int k = 0; k < 3; k++) p3d[3 * i + k] = F->mvStereo3Dpoints[i][k]; }
}

// Tracking::SearchLocalPoints' two steps on the reference's own code (src/Tracking.cc:4009-4067): Frame::SetPose, Frame::isInFrustum
// (src/Frame.cc:667-773, with the stand-in MapPoint's PredictScale restated from src/MapPoint.cc:688-731) for M points, then - when `assigned`
// is given - ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints) (src/ORBmatcher.cc:45-167).
// track: 7 arrays of M entries (in_view, proj_x, proj_y, proj_xr, depth, view_cos, scale_level as floats).
int ref_frame_search_local_points(void* h, const float* R, const float* t, int M, const float* pos, const float* normal, const float* min_dist, const float* max_dist,
                                  const uint8_t* bad, const uint8_t* has_obs, const uint8_t* desc, float cos_limit, float* track, int do_search, float th, int far_points,
                                  float th_far, float nnratio, int* assigned) {
    Frame* F = ((Holder*)h)->frame;
    const Sophus::SE3f Tcw = se3_from(R, t);
    F->SetPose(Tcw);
    std::vector<MapPoint> mps(M);
    std::vector<MapPoint*> vp(M);
    for (int i = 0; i < M; i++) {
        MapPoint& p = mps[i];
        p.pos = Eigen::Vector3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]); p.normal = Eigen::Vector3f(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
        p.minDist = min_dist[i]; p.maxDist = max_dist[i]; p.bad = bad && bad[i]; p.nObs = has_obs ? (has_obs[i] ? 2 : 0) : 1;
        p.descriptor = cv::Mat(1, 32, CV_8U); memcpy(p.descriptor.ptr(0), desc + 32 * (size_t)i, 32);
        vp[i] = &p;
        const bool in = F->isInFrustum(&p, cos_limit);
        track[i] = in && p.mbTrackInView ? 1.0f : 0.0f; track[M + i] = p.mTrackProjX; track[2 * (size_t)M + i] = p.mTrackProjY; track[3 * (size_t)M + i] = p.mTrackProjXR;
        track[4 * (size_t)M + i] = p.mTrackDepth; track[5 * (size_t)M + i] = p.mTrackViewCos; track[6 * (size_t)M + i] = (float)p.mnTrackScaleLevel;
    }
    if (!do_search) return 0;
    std::fill(F->mvpMapPoints.begin(), F->mvpMapPoints.end(), (MapPoint*)nullptr);
    ORBmatcher matcher(nnratio);
    const int n = matcher.SearchByProjection(*F, vp, th, far_points != 0, th_far);
    for (int i = 0; i < F->N; i++) assigned[i] = F->mvpMapPoints[i] ? (int)(F->mvpMapPoints[i] - mps.data()) : -1;
    return n;
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:2196-2324), the
// search of Tracking::Relocalization, on the reference's own Frame and matcher.  CurrentFrame = the holder's frame at pose (R, t), keypoints with
// occupied[i] != 0 hold a map point beforehand.  pKF = a key frame whose NK features carry the given map points (kind[i]: 0 none, 1 good, 2 bad,
// 3 good but in sAlreadyFound) and keypoint angles.  assigned[i] = index of the key-frame point written to CurrentFrame.mvpMapPoints[i], -1 untouched,
// -2 written and reset to NULL by the rotation check.
int ref_frame_search_keyframe(void* h, const float* R, const float* t, int NK, const float* pos, const uint8_t* kind, const float* min_dist, const float* max_dist,
                              const float* angle, const uint8_t* desc, float th, int orb_dist, int check_orientation, float nnratio, const uint8_t* occupied, int* assigned) {
    Frame* F = ((Holder*)h)->frame;
    F->SetPose(se3_from(R, t));
    KeyFrame KF;
    KF.N = NK; KF.mvKeysUn.assign(NK, cv::KeyPoint()); KF.mvpMapPoints.assign(NK, (MapPoint*)nullptr);
    std::vector<MapPoint> mps(NK);
    std::set<MapPoint*> found;
    for (int i = 0; i < NK; i++) {
        KF.mvKeysUn[i].angle = angle[i];
        MapPoint& p = mps[i];
        p.pos = Eigen::Vector3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]); p.minDist = min_dist[i]; p.maxDist = max_dist[i]; p.bad = kind[i] == 2; p.nObs = 1;
        p.descriptor = cv::Mat(1, 32, CV_8U); memcpy(p.descriptor.ptr(0), desc + 32 * (size_t)i, 32);
        if (kind[i]) KF.mvpMapPoints[i] = &p;
        if (kind[i] == 3) found.insert(&p);
    }
    MapPoint resident; resident.nObs = 0;          // any occupant blocks a keypoint in this search, observations or not
    std::vector<uint8_t> was(F->N, 0);
    auto reset = [&] { for (int i = 0; i < F->N; i++) { was[i] = occupied && occupied[i]; F->mvpMapPoints[i] = was[i] ? &resident : nullptr; } };
    reset();
    ORBmatcher matcher(nnratio, check_orientation != 0);
    const int n = matcher.SearchByProjection(*F, &KF, found, th, orb_dist);
    for (int i = 0; i < F->N; i++) { MapPoint* p = F->mvpMapPoints[i]; assigned[i] = (p && p != &resident) ? (int)(p - mps.data()) : -1; }
    if (check_orientation) {                        // keypoints that received a point and lost it to the rotation check: the same search without the check has them
        reset();
        ORBmatcher plain(nnratio, false);
        plain.SearchByProjection(*F, &KF, found, th, orb_dist);
        for (int i = 0; i < F->N; i++) { MapPoint* p = F->mvpMapPoints[i]; if (p && p != &resident && assigned[i] == -1) assigned[i] = -2; }
    }
    std::fill(F->mvpMapPoints.begin(), F->mvpMapPoints.end(), (MapPoint*)nullptr);
    return n;
}

// The checker's Sophus stand-in (oracle/slam_shim/sophus_model.h) evaluated on one pose (R, t) [+ a similarity (s, R2, t2)] and one point, for
// tests/test_sophus_action.py, which holds the Python host mirror (orb_slam3_detailed_comments_amd/sophus.py) against it.
// out (48 floats): unit quaternion 4 | rotationMatrix 9 | T * p 3 | inverse: quaternion 4, translation 3 | (T * T) quaternion 4, translation 3 |
// Sim3: quaternion 4, scale 1, rotationMatrix 9 (first 3 rows... all 9) -> see the layout in the test.
void ref_sophus_probe(const float* R, const float* t, const float* p, float s, const float* R2, const float* t2, float* out) {
    const Sophus::SE3f T = se3_from(R, t);
    const Eigen::Vector3f P(p[0], p[1], p[2]);
    int o = 0;
    auto putq = [&](const Eigen::Quaternionf& q) { out[o++] = q.x(); out[o++] = q.y(); out[o++] = q.z(); out[o++] = q.w(); };
    auto putv = [&](const Eigen::Vector3f& v) { for (int i = 0; i < 3; i++) out[o++] = v(i); };
    auto putm = [&](const Eigen::Matrix3f& m) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out[o++] = m(i, j); };
    putq(T.unit_quaternion()); putm(T.rotationMatrix()); putv(T * P);
    const Sophus::SE3f Ti = T.inverse(); putq(Ti.unit_quaternion()); putv(Ti.translation());
    const Sophus::SE3f TT = T * T; putq(TT.unit_quaternion()); putv(TT.translation());
    Eigen::Matrix3f Rm; Eigen::Vector3f tv;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rm(i, j) = R2[3 * i + j]; tv(i) = t2[i]; }
    const Sophus::Sim3f S(Sophus::RxSO3f(s, Rm), tv);
    putq(S.quaternion()); out[o++] = S.scale(); putm(S.rotationMatrix()); putv(S * P);
    const Sophus::Sim3f Si = S.inverse(); putq(Si.quaternion()); putv(Si.translation()); putv(Si * P);
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBmatcher.cc:1950-2184), the search of
// Tracking::TrackWithMotionModel, on the reference's own Frame and matcher.  CurrentFrame = the holder's frame at pose (R, t), keypoints with
// occupied[i] != 0 hold a map point with observations beforehand.  LastFrame = a copy of it (the reference's copy constructor) at pose (Rl, tl)
// whose NL keypoints carry the given octave / angle and whose map points are the NL given (valid[i] == 0: no map point).
// fwd_bwd[0..1] = bForward / bBackward as the reference derives them from the two poses (:1966-1975), recomputed here with the same
// expressions for the caller of the batched device search, which takes them as flags.  assigned[i] = index of the last-frame point written to
// CurrentFrame.mvpMapPoints[i], -1 untouched (or pre-occupied), -2 written and reset to NULL by the rotation check.
int ref_frame_search_lastframe(void* h, const float* R, const float* t, const float* Rl, const float* tl, int NL, const float* pos, const uint8_t* valid, const int* octave,
                               const float* angle, const uint8_t* has_obs, const uint8_t* desc, float th, int mono, int check_orientation, float nnratio, const uint8_t* occupied,
                               int* assigned, int* fwd_bwd) {
    Frame* F = ((Holder*)h)->frame;
    F->SetPose(se3_from(R, t));
    Frame Last(*F);
    Last.SetPose(se3_from(Rl, tl));
    Last.N = NL;
    Last.mvKeys.assign(NL, cv::KeyPoint()); Last.mvKeysUn.assign(NL, cv::KeyPoint());
    Last.mvpMapPoints.assign(NL, (MapPoint*)nullptr); Last.mvbOutlier.assign(NL, false);
    std::vector<MapPoint> mps(NL);
    for (int i = 0; i < NL; i++) {
        Last.mvKeys[i].octave = octave[i]; Last.mvKeysUn[i].octave = octave[i]; Last.mvKeys[i].angle = angle[i]; Last.mvKeysUn[i].angle = angle[i];
        MapPoint& p = mps[i];
        p.pos = Eigen::Vector3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]); p.nObs = has_obs ? (has_obs[i] ? 2 : 0) : 1;
        p.descriptor = cv::Mat(1, 32, CV_8U); memcpy(p.descriptor.ptr(0), desc + 32 * (size_t)i, 32);
        if (valid[i]) Last.mvpMapPoints[i] = &p;
    }
    MapPoint resident; resident.nObs = 3;
    std::vector<uint8_t> was(F->N, 0);
    for (int i = 0; i < F->N; i++) { was[i] = occupied && occupied[i]; F->mvpMapPoints[i] = was[i] ? &resident : nullptr; }
    {
        const Sophus::SE3f Tcw = F->GetPose();
        const Eigen::Vector3f twc = Tcw.inverse().translation();
        const Sophus::SE3f Tlw = Last.GetPose();
        const Eigen::Vector3f tlc = Tlw * twc;
        fwd_bwd[0] = tlc(2) > F->mb && !mono; fwd_bwd[1] = -tlc(2) > F->mb && !mono;
    }
    // a monocular frame has no right coordinates (mvuRight = -1, src/Frame.cc:360): the holder's frame is a stereo one, so hide them for bMono
    const std::vector<float> saved_u_right = F->mvuRight;
    if (mono) std::fill(F->mvuRight.begin(), F->mvuRight.end(), -1.0f);
    ORBmatcher matcher(nnratio, check_orientation != 0);
    const int n = matcher.SearchByProjection(*F, Last, th, mono != 0);
    for (int i = 0; i < F->N; i++) {
        MapPoint* p = F->mvpMapPoints[i];
        assigned[i] = (p && p != &resident) ? (int)(p - mps.data()) : -1;
    }
    // a keypoint that was free, received a point and lost it again to the rotation check is NULL afterwards, like one never touched: tell them
    // apart by running the search once more without the check
    if (check_orientation) {
        for (int i = 0; i < F->N; i++) F->mvpMapPoints[i] = was[i] ? &resident : nullptr;
        ORBmatcher plain(nnratio, false);
        plain.SearchByProjection(*F, Last, th, mono != 0);
        for (int i = 0; i < F->N; i++) {
            MapPoint* p = F->mvpMapPoints[i];
            if (p && p != &resident && assigned[i] == -1) assigned[i] = -2;
        }
    }
    std::fill(F->mvpMapPoints.begin(), F->mvpMapPoints.end(), (MapPoint*)nullptr);
    F->mvuRight = saved_u_right;
    return n;
}

// The same for a two-camera rig frame (ref_frame_fisheye): Frame::isInFrustum takes its Nleft != -1 branch (src/Frame.cc:754-766 ->
// isInFrustumChecks :1592-1650 once per camera) and SearchByProjection its right-camera branch (src/ORBmatcher.cc:170-236).
// track: 13 arrays of M entries: in_view, proj_x, proj_y, depth, view_cos, scale_level, in_view_r, proj_xr, proj_yr, depth_r, view_cos_r,
// scale_level_r, (unused).  The fields of a camera whose checks fail keep the sentinels the driver wrote (-7).
// pose_out (45 floats): mRcw 9, mtcw 3, mOw 3, mRwc 9, Trl rotation 9, Trl translation 3, Tlr translation 3, then 6 unused - what the product's
// OrbmFrustumRigView takes, read from the reference Frame itself after SetPose.
int ref_frame_search_local_points_rig(void* h, const float* R, const float* t, int M, const float* pos, const float* normal, const float* min_dist, const float* max_dist,
                                      const uint8_t* bad, const uint8_t* has_obs, const uint8_t* desc, float cos_limit, float* track, int do_search, float th, int far_points,
                                      float th_far, float nnratio, int* assigned, float* pose_out) {
    Frame* F = ((Holder*)h)->frame;
    const Sophus::SE3f Tcw = se3_from(R, t);
    F->SetPose(Tcw);
    {
        const Eigen::Matrix3f Rcw = F->GetPose().rotationMatrix(), Rwc = F->GetRwc(), Rrl = F->GetRelativePoseTrl().rotationMatrix();
        const Eigen::Vector3f tcw = F->GetPose().translation(), Ow = F->GetOw(), trl = F->GetRelativePoseTrl().translation(), tlr = F->GetRelativePoseTlr().translation();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { pose_out[3 * i + j] = Rcw(i, j); pose_out[15 + 3 * i + j] = Rwc(i, j); pose_out[24 + 3 * i + j] = Rrl(i, j); }
        for (int i = 0; i < 3; i++) { pose_out[9 + i] = tcw(i); pose_out[12 + i] = Ow(i); pose_out[33 + i] = trl(i); pose_out[36 + i] = tlr(i); }
    }
    std::vector<MapPoint> mps(M);
    std::vector<MapPoint*> vp(M);
    const size_t Ms = (size_t)M;
    for (int i = 0; i < M; i++) {
        MapPoint& p = mps[i];
        p.pos = Eigen::Vector3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]); p.normal = Eigen::Vector3f(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
        p.minDist = min_dist[i]; p.maxDist = max_dist[i]; p.bad = bad && bad[i]; p.nObs = has_obs ? (has_obs[i] ? 2 : 0) : 1;
        p.descriptor = cv::Mat(1, 32, CV_8U); memcpy(p.descriptor.ptr(0), desc + 32 * (size_t)i, 32);
        p.mTrackProjX = p.mTrackProjY = p.mTrackDepth = p.mTrackViewCos = p.mTrackProjXR = p.mTrackProjYR = p.mTrackDepthR = p.mTrackViewCosR = -7.0f;
        vp[i] = &p;
        F->isInFrustum(&p, cos_limit);
        track[i] = p.mbTrackInView; track[Ms + i] = p.mTrackProjX; track[2 * Ms + i] = p.mTrackProjY; track[3 * Ms + i] = p.mTrackDepth; track[4 * Ms + i] = p.mTrackViewCos;
        track[5 * Ms + i] = (float)p.mnTrackScaleLevel;
        track[6 * Ms + i] = p.mbTrackInViewR; track[7 * Ms + i] = p.mTrackProjXR; track[8 * Ms + i] = p.mTrackProjYR; track[9 * Ms + i] = p.mTrackDepthR; track[10 * Ms + i] = p.mTrackViewCosR;
        track[11 * Ms + i] = (float)p.mnTrackScaleLevelR;
    }
    if (!do_search) return 0;
    std::fill(F->mvpMapPoints.begin(), F->mvpMapPoints.end(), (MapPoint*)nullptr);
    ORBmatcher matcher(nnratio);
    const int n = matcher.SearchByProjection(*F, vp, th, far_points != 0, th_far);
    for (int i = 0; i < F->N; i++) assigned[i] = F->mvpMapPoints[i] ? (int)(F->mvpMapPoints[i] - mps.data()) : -1;
    return n;
}

// bench.py's cpu_baseline: the reference's steady state - two long-lived extractors (Tracking owns them), one Frame temporary per stereo
// pair built in the same storage (the constructor itself runs the two extractions on two threads).  Returns the number of frames
// constructed in `seconds`; *elapsed = the time they took.
int ref_frame_stereo_repeat(const uint8_t* L, const uint8_t* R, int w, int h, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th,
                            float fx, float fy, float cx, float cy, float bf, float th_depth, double seconds, double* elapsed, int* matches, double* stage_ms) {
    ORBextractor left(nfeatures, scale_factor, nlevels, ini_th, min_th), right(nfeatures, scale_factor, nlevels, ini_th, min_th);
    Pinhole cam(std::vector<float>{fx, fy, cx, cy});
    cv::Mat imL(h, w, CV_8UC1, (void*)L, (size_t)w), imR(h, w, CV_8UC1, (void*)R, (size_t)w);
    cv::Mat K = cam.toK();
    cv::Mat dist(4, 1, CV_32F); for (int i = 0; i < 4; i++) dist.at<float>(i) = 0.0f;
    void* storage = ::operator new(sizeof(Frame));
    memset(storage, 0, sizeof(Frame));
    reinterpret_cast<Frame*>(storage)->mb = bf / fx;
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0; double dt = 0;
    do {
        Frame* F = new (storage) Frame(imL, imR, 0.0, &left, &right, nullptr, K, dist, bf, th_depth, &cam);
        int m = 0; for (int i = 0; i < F->N; i++) m += F->mvuRight[i] >= 0;
        *matches = m;
#ifdef REGISTER_TIMES                          // the reference's own timers (src/Frame.cc:132-146 "ORB Extraction", :158-170 "Stereo Matching")
        if (stage_ms) { stage_ms[0] += F->mTimeORB_Ext; stage_ms[1] += F->mTimeStereoMatch; }
#endif
        const float mb = F->mb;
        F->~Frame();
        reinterpret_cast<Frame*>(storage)->mb = mb;
        n++;
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (dt < seconds);
    ::operator delete(storage);
    *elapsed = dt;
    return n;
}

// mvKeys, mvKeysUn, mDescriptors, mvuRight, mvDepth (N entries each) and mvKeysRight, mDescriptorsRight
void ref_frame_get(void* h, void* keys, void* keys_un, uint8_t* desc, float* u_right, float* depth, void* keys_right, uint8_t* desc_right) {
    Frame* F = ((Holder*)h)->frame;
    put_keys(F->mvKeys, keys); put_keys(F->mvKeysUn, keys_un); put_keys(F->mvKeysRight, keys_right);
    for (int i = 0; i < F->N; i++) { memcpy(desc + 32 * (size_t)i, F->mDescriptors.ptr(i), 32); u_right[i] = F->mvuRight[i]; depth[i] = F->mvDepth[i]; }
    for (size_t i = 0; i < F->mvKeysRight.size(); i++) memcpy(desc_right + 32 * i, F->mDescriptorsRight.ptr((int)i), 32);
}
// out = {mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv, mbf, mb}
void ref_frame_constants(void* h, float* out) {
    Frame* F = ((Holder*)h)->frame;
    out[0] = Frame::mnMinX; out[1] = Frame::mnMinY; out[2] = Frame::mnMaxX; out[3] = Frame::mnMaxY;
    out[4] = Frame::mfGridElementWidthInv; out[5] = Frame::mfGridElementHeightInv; out[6] = F->mbf; out[7] = F->mb;
}
int ref_frame_features_in_area(void* h, float x, float y, float r, int min_level, int max_level, int* idx, int cap) {
    const std::vector<size_t> v = ((Holder*)h)->frame->GetFeaturesInArea(x, y, r, min_level, max_level);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) idx[i] = (int)v[i];
    return (int)v.size();
}

}  // extern "C"
