// loopclosing_world.h — TEST INFRASTRUCTURE (part of oracle/; never shipped).  Force-included after full_world.h to COMPILE the reference's
// src/LoopClosing.cc - unmodified and in place - against the drop-in ORBmatcher.h (oracle/Makefile, _ref/loopclosing_dropin.o): the caller of
// SearchByBoW(pKF, pKF) (:845), SearchByProjection with a Sim3 (:703, :1082, :1216-1240), SearchBySim3 and both Fuse(Sim3) overloads (:2700-2765).
// Compile-only: g2o's Sim3, the double-precision Eigen / Sophus types, the Sim3 solver, the optimiser, Atlas, Tracking, LocalMapping are DECLARED here;
// nothing is linked or run.
#ifndef ORBX_LOOPCLOSING_WORLD_H
#define ORBX_LOOPCLOSING_WORLD_H
#define ATLAS_H
#define LOCALMAPPING_H
#define TRACKING_H
#define SYSTEM_H
#define OPTIMIZER_H
#define G2OTYPES_H
#define SIM3SOLVER_H
#define ORB_SLAM3_SETTINGS_H
#define G2O_SEVEN_DOF_EXPMAP_TYPES
#include <list>
#include <map>
#include <set>
#include <thread>
#include <unistd.h>
#include "KeyFrame.h"
#include "Frame.h"
namespace Eigen {
template <class T> using aligned_allocator = std::allocator<T>;
struct Matrix3d;
struct Quaterniond {
    Quaterniond(); explicit Quaterniond(const Matrix3d& R); Quaterniond(double w, double x, double y, double z);
    Matrix3d toRotationMatrix() const; Quaterniond operator*(const Quaterniond& o) const; Vector3d operator*(const Vector3d& v) const;
    template <class T> Quaternionf cast() const; Quaterniond inverse() const; Quaterniond conjugate() const; void normalize();
    double w() const, x() const, y() const, z() const;
};
struct Matrix3d {
    Matrix3d(); Matrix3d(const Quaterniond& q);
    static Matrix3d Identity(); Matrix3d transpose() const; Matrix3d operator*(const Matrix3d& o) const; Vector3d operator*(const Vector3d& v) const;
    template <class T> typename LmCastM3<T>::type cast() const; double& operator()(int r, int c); double operator()(int r, int c) const;
};
template <> struct Quaternionf::CastQ<double> { typedef Quaterniond type; };
template <> struct Quaternionf::CastQ<float> { typedef Quaternionf type; };
struct MatrixXd { static MatrixXd Zero(int r, int c); };
struct VectorXd {};
template <> struct LmCastV3<double> { typedef Vector3d type; };
template <> struct LmCastV3<float> { typedef Vector3f type; };
template <> struct LmCastM3<double> { typedef Matrix3d type; };
template <> struct LmCastM3<float> { typedef Matrix3f type; };
}
namespace g2o {
class Sim3 {
public:
    Sim3(); Sim3(const Eigen::Quaterniond& r, const Eigen::Vector3d& t, double s); Sim3(const Eigen::Matrix3d& R, const Eigen::Vector3d& t, double s);
    const Eigen::Quaterniond& rotation() const; const Eigen::Vector3d& translation() const; const double& scale() const;
    Sim3 inverse() const; Sim3 operator*(const Sim3& o) const; Eigen::Vector3d map(const Eigen::Vector3d& xyz) const;
};
}
namespace ORB_SLAM3 {
Eigen::Vector3d LogSO3(const Eigen::Matrix3d& R);
Eigen::Matrix3d ExpSO3(const Eigen::Vector3d& w);
class Verbose {
public:
    enum eLevel { VERBOSITY_QUIET = 0, VERBOSITY_NORMAL = 1, VERBOSITY_VERBOSE = 2, VERBOSITY_VERY_VERBOSE = 3, VERBOSITY_DEBUG = 4 };
    static eLevel th;
    static void PrintMess(std::string str, eLevel lev);
};
class System { public: enum eSensor { MONOCULAR = 0, STEREO = 1, RGBD = 2, IMU_MONOCULAR = 3, IMU_STEREO = 4, IMU_RGBD = 5 }; };
class Viewer;
class Atlas {
public:
    Map* GetCurrentMap(); void SetMapBad(Map* pMap); int CountMaps(); void RemoveBadMaps(); void InformNewBigChange(); unsigned long int GetLastInitKFid(); void ChangeMap(Map* pMap);
};
class LocalMapping {
public:
    void Release(); bool isStopped(); void RequestStop(); void EmptyQueue(); bool isFinished();
    std::mutex mMutexImuInit; bool mbBadImu; double mScale;
};
class Tracking {
public:
    int mSensor; KeyFrame* GetLastKeyFrame(); void UpdateFrameIMU(const float s, const IMU::Bias& b, KeyFrame* pCurrentKeyFrame); void SetStepByStep(bool bSet);
};
class Sim3Solver {
public:
    Sim3Solver(KeyFrame* pKF1, KeyFrame* pKF2, const std::vector<MapPoint*>& vpMatched12, const bool bFixScale = true,
               const std::vector<KeyFrame*> vpKeyFrameMatchedMP = std::vector<KeyFrame*>());
    void SetRansacParameters(double probability = 0.99, int minInliers = 6, int maxIterations = 300);
    Eigen::Matrix4f iterate(int nIterations, bool& bNoMore, std::vector<bool>& vbInliers, int& nInliers);
    Eigen::Matrix4f iterate(int nIterations, bool& bNoMore, std::vector<bool>& vbInliers, int& nInliers, bool& bConverge);
    Eigen::Matrix3f GetEstimatedRotation(); Eigen::Vector3f GetEstimatedTranslation(); float GetEstimatedScale();
};
}
#include "LoopClosing.h"
namespace ORB_SLAM3 {
class Optimizer {
public:
    static void GlobalBundleAdjustemnt(Map* pMap, int nIterations = 5, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
    static void FullInertialBA(Map* pMap, int its, const bool bFixLocal = false, const unsigned long nLoopKF = 0, bool* pbStopFlag = NULL, bool bInit = false, float priorG = 1e2,
                               float priorA = 1e6, Eigen::VectorXd* vSingVal = NULL, bool* bHess = NULL);
    static void OptimizeEssentialGraph(Map* pMap, KeyFrame* pLoopKF, KeyFrame* pCurKF, const LoopClosing::KeyFrameAndPose& NonCorrectedSim3,
                                       const LoopClosing::KeyFrameAndPose& CorrectedSim3, const std::map<KeyFrame*, std::set<KeyFrame*> >& LoopConnections, const bool& bFixScale);
    static void OptimizeEssentialGraph(KeyFrame* pCurKF, std::vector<KeyFrame*>& vpFixedKFs, std::vector<KeyFrame*>& vpFixedCorrectedKFs, std::vector<KeyFrame*>& vpNonFixedKFs,
                                       std::vector<MapPoint*>& vpNonCorrectedMPs);
    static void OptimizeEssentialGraph4DoF(Map* pMap, KeyFrame* pLoopKF, KeyFrame* pCurKF, const LoopClosing::KeyFrameAndPose& NonCorrectedSim3,
                                           const LoopClosing::KeyFrameAndPose& CorrectedSim3, const std::map<KeyFrame*, std::set<KeyFrame*> >& LoopConnections);
    static int OptimizeSim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches1, g2o::Sim3& g2oS12, const float th2, const bool bFixScale,
                            Eigen::Matrix<double, 7, 7>& mAcumHessian, const bool bAllPoints = false);
    static void MergeInertialBA(KeyFrame* pCurrKF, KeyFrame* pMergeKF, bool* pbStopFlag, Map* pMap, LoopClosing::KeyFrameAndPose& corrPoses);
    static void LocalBundleAdjustment(KeyFrame* pMainKF, std::vector<KeyFrame*> vpAdjustKF, std::vector<KeyFrame*> vpFixedKF, bool* pbStopFlag);
    static void InertialOptimization(Map* pMap, Eigen::Vector3d& bg, Eigen::Vector3d& ba, float priorG = 1e2, float priorA = 1e6);
};
}
#endif
