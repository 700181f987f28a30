// slam_types.h — TEST INFRASTRUCTURE (part of oracle/; never shipped).  Stand-ins for Eigen / Sophus and for the MapPoint and KeyFrame classes
// of the reference, shared by slam_world.h (the world the reference's ORBmatcher.cc is compiled over) and frame_world.h (the world the
// reference's Frame.cc is compiled over).  See slam_world.h for what is restated here and why.  The CAMERA MODELS are the reference's own:
// include/CameraModels/{GeometricCamera,Pinhole,KannalaBrandt8}.h are included from here, and src/CameraModels/Pinhole.cpp + KannalaBrandt8.cpp
// are compiled unmodified into every _ref library (oracle/Makefile, CAMSRC) over the stand-in Eigen of this file and eigen_small.h.
#ifndef ORBX_SLAM_TYPES_H
#define ORBX_SLAM_TYPES_H

#ifndef ORBX_REAL_MAPPOINT
#define MAPPOINT_H
#endif
#ifndef ORBX_REAL_KEYFRAME
#define KEYFRAME_H
#endif

#include <cmath>
#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <vector>
#include <opencv2/core/core.hpp>
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

using namespace std;     // the reference's headers rely on it (include/ORBmatcher.h:78 uses an unqualified pair<>)

namespace Eigen {
struct Vector2f {
    float d[2];
    Vector2f() : d{0, 0} {}
    Vector2f(float x, float y) : d{x, y} {}
    float& operator()(int i) { return d[i]; }
    const float& operator()(int i) const { return d[i]; }
    float& operator[](int i) { return d[i]; }
    const float& operator[](int i) const { return d[i]; }
};
#ifdef ORBX_LOCALMAPPING_WORLD     // (localmapping_world.h: declarations for the compile check of LocalMapping.cc; nothing of it is linked)
template <class T> struct LmCastV3;
template <class T> struct LmCastM3;
#endif
struct Vector3f {
    float d[3];
#ifdef ORBX_LOCALMAPPING_WORLD
    Vector3f cross(const Vector3f& o) const;
    Vector3f& operator-=(const Vector3f& o);
    Vector3f& operator*=(float s);
    struct CommaInitV3 { CommaInitV3& operator,(float v); };
    CommaInitV3 operator<<(float v);
    template <class T> typename LmCastV3<T>::type cast() const;
#endif
    Vector3f() : d{0, 0, 0} {}
    Vector3f(float x, float y, float z) : d{x, y, z} {}
    float& operator()(int i) { return d[i]; }
    const float& operator()(int i) const { return d[i]; }
    float& operator[](int i) { return d[i]; }
    const float& operator[](int i) const { return d[i]; }
    void setZero() { d[0] = d[1] = d[2] = 0; }
    static Vector3f Zero() { return Vector3f(); }
    float* data() { return d; }
    size_t size() const { return 3; }
    // Eigen >= 3.3 (required by the vendored Sophus, Thirdparty/Sophus/CMakeLists.txt:35): a fixed-size reduction of three terms is unrolled by
    // redux_novec_unroller, which halves the range (Eigen/src/Core/Redux.h): a0 + (a1 + a2).  dot(), norm(), trace() and every coefficient of a
    // 3x3 product (ProductEvaluators.h: (lhs.row(i).transpose().cwiseProduct(rhs.col(j))).sum()) are such reductions.
    float dot(const Vector3f& o) const { return d[0] * o.d[0] + (d[1] * o.d[1] + d[2] * o.d[2]); }
    float norm() const { return std::sqrt(dot(*this)); }
};
inline Vector3f operator-(const Vector3f& a, const Vector3f& b) { return Vector3f(a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]); }
inline Vector3f operator+(const Vector3f& a, const Vector3f& b) { return Vector3f(a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]); }
inline Vector3f operator-(const Vector3f& a) { return Vector3f(-a.d[0], -a.d[1], -a.d[2]); }
inline Vector3f operator/(const Vector3f& a, float s) { return Vector3f(a.d[0] / s, a.d[1] / s, a.d[2] / s); }
inline Vector3f operator*(const Vector3f& a, float s) { return Vector3f(a.d[0] * s, a.d[1] * s, a.d[2] * s); }
inline Vector3f operator*(float s, const Vector3f& a) { return a * s; }
struct Matrix3f {
    float m[3][3];
#ifdef ORBX_LOCALMAPPING_WORLD
    template <class T> typename LmCastM3<T>::type cast() const;
    void setIdentity();
#endif
    Matrix3f() : m{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}} {}
    static Matrix3f Identity() { Matrix3f r; r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0f; return r; }
    static Matrix3f Zero() { return Matrix3f(); }
    float& operator()(int r, int c) { return m[r][c]; }
    const float& operator()(int r, int c) const { return m[r][c]; }
    float* data() { return &m[0][0]; }
    size_t size() const { return 9; }
    Vector3f row(int r) const { return Vector3f(m[r][0], m[r][1], m[r][2]); }
    Matrix3f transpose() const { Matrix3f r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = m[j][i]; return r; }
    // Eigen/src/LU/InverseImpl.h, compute_inverse<MatrixType, ResultType, 3>: cofactor_3x3<i,j> = m(i1,j1) m(i2,j2) - m(i1,j2) m(i2,j1) with
    // i1 = (i+1)%3, i2 = (i+2)%3 (likewise j); det = (cofactors_col0.cwiseProduct(matrix.col(0))).sum(); result(i,j) = cofactor<j,i> * (1 / det)
    float cofactor(int i, int j) const { const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1]; }
    Matrix3f inverse() const {
        Matrix3f r;
        const float c0 = cofactor(0, 0), c1 = cofactor(1, 0), c2 = cofactor(2, 0);
        const float det = c0 * m[0][0] + (c1 * m[1][0] + c2 * m[2][0]), invdet = 1.0f / det;
        r.m[0][0] = c0 * invdet; r.m[0][1] = c1 * invdet; r.m[0][2] = c2 * invdet;
        for (int i = 1; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = cofactor(j, i) * invdet;
        return r;
    }
};
inline Vector3f operator*(const Matrix3f& A, const Vector3f& v) {
    return Vector3f(A.m[0][0] * v.d[0] + (A.m[0][1] * v.d[1] + A.m[0][2] * v.d[2]), A.m[1][0] * v.d[0] + (A.m[1][1] * v.d[1] + A.m[1][2] * v.d[2]),
                    A.m[2][0] * v.d[0] + (A.m[2][1] * v.d[1] + A.m[2][2] * v.d[2]));
}
inline Matrix3f operator*(const Matrix3f& A, const Matrix3f& B) {
    Matrix3f r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = A.m[i][0] * B.m[0][j] + (A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j]);
    return r;
}
inline Matrix3f operator-(const Matrix3f& A) { Matrix3f r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = -A.m[i][j]; return r; }
inline Matrix3f operator*(const Matrix3f& A, float s) { Matrix3f r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = A.m[i][j] * s; return r; }
}  // namespace Eigen

#include "eigen_small.h"       // what the reference's camera models need beyond the types above: Matrix<float,3,4>, Matrix4f, JacobiSVD<Matrix4f>, ...
#include "sophus_model.h"     // Sophus::SO3f / SE3f / RxSO3f / Sim3f: the vendored Sophus restated over a stand-in Eigen::Quaternionf

// The reference's own camera models.  GeometricCamera.h pulls in Converter.h and GeometricTools.h (g2o, Eigen/Dense: unbuildable here, and nothing
// of them is used by the camera models); Pinhole.h / KannalaBrandt8.h name TwoViewReconstruction (monocular initialisation, out of scope,
// SURVEY.md 8): a two-method stand-in that is never called.
#ifndef CONVERTER_H
#define CONVERTER_H
#endif
#define GEOMETRIC_TOOLS_H
#define TwoViewReconstruction_H
namespace ORB_SLAM3 {
class TwoViewReconstruction {
public:
    TwoViewReconstruction(const Eigen::Matrix3f&) {}
    bool Reconstruct(const std::vector<cv::KeyPoint>&, const std::vector<cv::KeyPoint>&, const std::vector<int>&, Sophus::SE3f&, std::vector<cv::Point3f>&, std::vector<bool>&) { return false; }
};
}
#include "CameraModels/GeometricCamera.h"
#include "CameraModels/Pinhole.h"
#include "CameraModels/KannalaBrandt8.h"

namespace ORB_SLAM3 {

class KeyFrame;
class Frame;
class MapPoint;
class Map;

}  // namespace ORB_SLAM3

namespace ORB_SLAM3 {

#ifndef ORBX_REAL_MAPPOINT
class MapPoint {
public:
    // tracking fields (include/MapPoint.h:171-179)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackDepthR = 0, mTrackProjXR = 0, mTrackProjYR = 0;
    bool mbTrackInView = false, mbTrackInViewR = false;
    int mnTrackScaleLevel = 0, mnTrackScaleLevelR = -1;
    float mTrackViewCos = 1, mTrackViewCosR = 1;
    // state behind the accessors
    int id = -1;
    bool bad = false;
    int nObs = 1;
    Eigen::Vector3f pos, normal;
    float minDist = 0, maxDist = 1e9f;
    cv::Mat descriptor;
    std::map<KeyFrame*, std::tuple<int, int>> observations;
    MapPoint* replacedBy = nullptr;         // log of Replace()

    bool isBad() { return bad; }
    int Observations() { return nObs; }
    cv::Mat GetDescriptor() { return descriptor.clone(); }
    Eigen::Vector3f GetWorldPos() { return pos; }
    Eigen::Vector3f GetNormal() { return normal; }
    float GetMinDistanceInvariance() { return 0.8f * minDist; }           // src/MapPoint.cc:658-671
    float GetMaxDistanceInvariance() { return 1.2f * maxDist; }
    template <class T> int PredictScale(const float& currentDist, T* pF) {  // src/MapPoint.cc:688-731
        float ratio = maxDist / currentDist;
        // the reference's translation unit is `using namespace std`: log(float) is std::log(float) = logf, the division and ceil are float
        // (pinned against the reference's own MapPoint.cc by tests/test_models.py::test_predict_scale_uses_float_log)
        int nScale = std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
        return nScale;
    }
    bool IsInKeyFrame(KeyFrame* pKF) { return observations.count(pKF) != 0; }
    std::tuple<int, int> GetIndexInKeyFrame(KeyFrame* pKF) { auto it = observations.find(pKF); return it != observations.end() ? it->second : std::tuple<int, int>(-1, -1); }
    void AddObservation(KeyFrame* pKF, int idx) { observations[pKF] = std::tuple<int, int>(idx, -1); nObs++; }
    void Replace(MapPoint* pMP) { replacedBy = pMP; bad = true; }
};

#endif

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

// what Frame and KeyFrame share
class FeatureHolder {
public:
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    int mnScaleLevels = 0;
    float mfLogScaleFactor = 0;
    float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
    float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0;
    GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
    std::vector<size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS], mGridRight[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    Sophus::SE3f mTcw, mTrl;             // world -> (left) camera; left -> right camera

    bool PosInGrid(const cv::KeyPoint& kp, int& posX, int& posY) {       // src/Frame.cc:964-975
        posX = round((kp.pt.x - mnMinX) * mfGridElementWidthInv);
        posY = round((kp.pt.y - mnMinY) * mfGridElementHeightInv);
        if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) return false;
        return true;
    }
    // src/Frame.cc:469-503 (AssignFeaturesToGrid); nleft = Nleft / NLeft of the owner
    void AssignFeaturesToGrid(int nleft) {
        for (int i = 0; i < FRAME_GRID_COLS; i++) for (int j = 0; j < FRAME_GRID_ROWS; j++) { mGrid[i][j].clear(); mGridRight[i][j].clear(); }
        for (int i = 0; i < N; i++) {
            const cv::KeyPoint& kp = (nleft == -1) ? mvKeysUn[i] : (i < nleft) ? mvKeys[i] : mvKeysRight[i - nleft];
            int x, y;
            if (PosInGrid(kp, x, y)) {
                if (nleft == -1 || i < nleft) mGrid[x][y].push_back(i);
                else mGridRight[x][y].push_back(i - nleft);
            }
        }
    }
};

#ifndef ORBX_REAL_KEYFRAME
class KeyFrame : public FeatureHolder {
public:
    int NLeft = -1, NRight = -1;
    const int mnGridCols = FRAME_GRID_COLS, mnGridRows = FRAME_GRID_ROWS;
    std::vector<MapPoint*> mvpMapPoints;

    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
#ifndef ORBX_REAL_MAPPOINT
    std::set<MapPoint*> GetMapPoints() { std::set<MapPoint*> s; for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p); return s; }   // src/KeyFrame.cc:370-385
#else
    std::set<MapPoint*> GetMapPoints();            // defined where the real MapPoint is complete (oracle/ref_mappoint_driver.cpp)
#endif
    void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
    // what src/MapPoint.cc calls on a key frame
    long unsigned int mnId = 0, mnFrameId = 0;
    bool mbBadKF = false;
    bool isBad() { return mbBadKF; }
    Map* GetMap() { return nullptr; }
    void EraseMapPointMatch(const int& idx) { mvpMapPoints[idx] = nullptr; }
    void EraseMapPointMatch(MapPoint* pMP) { for (auto& p : mvpMapPoints) if (p == pMP) p = nullptr; }
    void ReplaceMapPointMatch(const int& idx, MapPoint* pMP) { mvpMapPoints[idx] = pMP; }
    Sophus::SE3f GetPose() { return mTcw; }
    Sophus::SE3f GetPoseInverse() { return mTcw.inverse(); }
    Eigen::Vector3f GetCameraCenter() { return mTcw.inverse().translation(); }
    Sophus::SE3f GetRightPose() { return mTrl * mTcw; }
    Sophus::SE3f GetRightPoseInverse() { return (mTrl * mTcw).inverse(); }
    Eigen::Vector3f GetRightCameraCenter() { return (mTrl * mTcw).inverse().translation(); }
    bool IsInImage(const float& x, const float& y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }   // src/KeyFrame.cc:894-897
    std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const bool bRight = false) const {   // src/KeyFrame.cc:843-891
        std::vector<size_t> vIndices;
        vIndices.reserve(N);
        float factorX = r, factorY = r;
        const int nMinCellX = max(0, (int)floor((x - mnMinX - factorX) * mfGridElementWidthInv));
        if (nMinCellX >= mnGridCols) return vIndices;
        const int nMaxCellX = min((int)mnGridCols - 1, (int)ceil((x - mnMinX + factorX) * mfGridElementWidthInv));
        if (nMaxCellX < 0) return vIndices;
        const int nMinCellY = max(0, (int)floor((y - mnMinY - factorY) * mfGridElementHeightInv));
        if (nMinCellY >= mnGridRows) return vIndices;
        const int nMaxCellY = min((int)mnGridRows - 1, (int)ceil((y - mnMinY + factorY) * mfGridElementHeightInv));
        if (nMaxCellY < 0) return vIndices;
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<size_t>& vCell = (!bRight) ? mGrid[ix][iy] : mGridRight[ix][iy];
                for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
                    const cv::KeyPoint& kpUn = (NLeft == -1) ? mvKeysUn[vCell[j]] : (!bRight) ? mvKeys[vCell[j]] : mvKeysRight[vCell[j]];
                    const float distx = kpUn.pt.x - x, disty = kpUn.pt.y - y;
                    if (fabs(distx) < r && fabs(disty) < r) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    }
};

#endif

}  // namespace ORB_SLAM3
#endif
