// tracking_world.h — TEST INFRASTRUCTURE (part of oracle/; never shipped).  Force-included after full_world.h to COMPILE the reference's src/Tracking.cc -
// unmodified and in place - against the drop-in ORBextractor.h and ORBmatcher.h (oracle/Makefile, _ref/tracking_dropin.o): the caller that constructs the
// extractors (Tracking.cc:631-635, :1328-1332) and calls SearchByBoW (:3183, :4371), SearchByProjection x3 (:3389, :4062, :4480-4500) and
// SearchForInitialization (:2875), over the reference's own Frame / KeyFrame / MapPoint headers.  Compile-only: everything else Tracking.cc talks to -
// System, Atlas, LocalMapping, LoopClosing, Viewer, the drawers, Settings, Optimizer, MLPnPsolver, the IMU preintegration - is DECLARED here (their headers
// pull in g2o, Pangolin and the rest of the system); nothing is linked or run.
#ifndef ORBX_TRACKING_WORLD_H
#define ORBX_TRACKING_WORLD_H
#define VIEWER_H
#define FRAMEDRAWER_H
#define MAPDRAWER_H
#define ATLAS_H
#define LOCALMAPPING_H
#define LOOPCLOSING_H
#define SYSTEM_H
#define ORB_SLAM3_SETTINGS_H
#define OPTIMIZER_H
#define G2OTYPES_H
#define ORB_SLAM3_MLPNPSOLVER_H
#include <list>
#include <set>
#include <thread>
#include <unordered_set>
#include <unistd.h>
#include "KeyFrame.h"
#include "Frame.h"
namespace ORB_SLAM3 {
class Tracking;
class Verbose {
public:
    enum eLevel { VERBOSITY_QUIET = 0, VERBOSITY_NORMAL = 1, VERBOSITY_VERBOSE = 2, VERBOSITY_VERY_VERBOSE = 3, VERBOSITY_DEBUG = 4 };
    static eLevel th;
    static void PrintMess(std::string str, eLevel lev);
};
class System {
public:
    enum eSensor { MONOCULAR = 0, STEREO = 1, RGBD = 2, IMU_MONOCULAR = 3, IMU_STEREO = 4, IMU_RGBD = 5 };
    void ResetActiveMap();
    void SaveTrajectoryEuRoC(const std::string& filename); void SaveTrajectoryEuRoC(const std::string& filename, Map* pMap);
    void SaveKeyFrameTrajectoryEuRoC(const std::string& filename);
    void SaveKeyFrameTrajectoryEuRoC(const std::string& filename, Map* pMap);
};
class Viewer { public: void RequestStop(); bool isStopped(); void Release(); bool both; };
class FrameDrawer { public: void Update(Tracking* pTracker); bool both; };
class MapDrawer { public: void SetCurrentCameraPose(const Sophus::SE3f& Tcw); };
class Atlas {
public:
    void CreateNewMap(); unsigned long int GetLastInitKFid(); void AddKeyFrame(KeyFrame* pKF); void AddMapPoint(MapPoint* pMP);
    GeometricCamera* AddCamera(GeometricCamera* pCam); std::vector<GeometricCamera*> GetAllCameras();
    void SetReferenceMapPoints(const std::vector<MapPoint*>& vpMPs); long unsigned int MapPointsInMap(); long unsigned KeyFramesInMap();
    std::vector<KeyFrame*> GetAllKeyFrames(); std::vector<MapPoint*> GetAllMapPoints(); std::vector<Map*> GetAllMaps();
    void clearMap(); void clearAtlas(); Map* GetCurrentMap(); bool isInertial(); void SetInertialSensor(); bool isImuInitialized();
};
class LocalMapping {
public:
    void InsertKeyFrame(KeyFrame* pKF); bool IsInitializing(); bool SetNotStop(bool flag); int KeyframesInQueue(); bool stopRequested(); bool isStopped();
    void RequestResetActiveMap(Map* pMap); void RequestReset(); void InterruptBA(); bool AcceptKeyFrames();
    int mnMatchesInliers; bool mbFarPoints, mbBadImu; float mThFarPoints; double mFirstTs;
    std::vector<double> vdKFInsert_ms, vdMPCulling_ms, vdMPCreation_ms, vdLBA_ms, vdKFCulling_ms, vdLMTotal_ms, vdLBASync_ms, vdKFCullingSync_ms;
    std::vector<int> vnLBA_edges, vnLBA_KFopt, vnLBA_KFfixed, vnLBA_MPs; int nLBA_exec, nLBA_abort;
};
class LoopClosing {
public:
    void RequestResetActiveMap(Map* pMap); void RequestReset();
    std::vector<double> vdDataQuery_ms, vdEstSim3_ms, vdPRTotal_ms, vdMergeMaps_ms, vdWeldingBA_ms, vdMergeOptEss_ms, vdMergeTotal_ms, vdLoopFusion_ms, vdLoopOptEss_ms, vdLoopTotal_ms,
                        vdGBA_ms, vdUpdateMap_ms, vdFGBATotal_ms;
    std::vector<int> vnMergeKFs, vnMergeMPs, vnLoopKFs, vnGBAKFs, vnGBAMPs; int nMerges, nLoop, nFGBA_exec, nFGBA_abort;
};
class Settings {
public:
    enum CameraType { PinHole = 0, Rectified = 1, KannalaBrandt = 2 };
    CameraType cameraType(); GeometricCamera* camera1(); GeometricCamera* camera2(); cv::Mat camera1DistortionCoef();
    Sophus::SE3f Tlr(); float bf(); float b(); float thDepth(); bool needToUndistort(); float fps(); bool rgb(); float noiseGyro(); float noiseAcc(); float gyroWalk();
    float accWalk(); float imuFrequency(); Sophus::SE3f Tbc(); bool insertKFsWhenLost(); float depthMapFactor(); int nFeatures(); int nLevels(); float initThFAST();
    float minThFAST(); float scaleFactor();
};
class Optimizer {
public:
    static void GlobalBundleAdjustemnt(Map* pMap, int nIterations = 5, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
    static int PoseOptimization(Frame* pFrame);
    static int PoseInertialOptimizationLastKeyFrame(Frame* pFrame, bool bRecInit = false);
    static int PoseInertialOptimizationLastFrame(Frame* pFrame, bool bRecInit = false);
};
class MLPnPsolver {
public:
    MLPnPsolver(const Frame& F, const std::vector<MapPoint*>& vpMapPointMatches);
    void SetRansacParameters(double probability = 0.99, int minInliers = 8, int maxIterations = 300, int minSet = 6, float epsilon = 0.4, float th2 = 5.991);
    bool iterate(int nIterations, bool& bNoMore, std::vector<bool>& vbInliers, int& nInliers, Eigen::Matrix4f& Tout);
};
}
#endif
