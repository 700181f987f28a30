// full_world.h — TEST INFRASTRUCTURE (part of oracle/; never shipped).  Force-included for the third matcher world
// (oracle/Makefile, _ref/libmw_ref_full.so and _ref/libmw_facade_full.so): ORBmatcher.cc / the facade run over the reference's OWN Frame,
// KeyFrame and MapPoint classes (their headers and Frame.cc, KeyFrame.cc, MapPoint.cc compiled unmodified and in place).  What remains a
// stand-in: Map and KeyFrameDatabase (bookkeeping KeyFrame.cc / MapPoint.cc call into), the IMU types, the serialization helpers named by
// never-instantiated serialize() templates, the pinhole camera and the Eigen / Sophus algebra (slam_types.h, frame_world.h).
#ifndef ORBX_FULL_WORLD_H
#define ORBX_FULL_WORLD_H
#define ORBX_REAL_MAPPOINT
#define ORBX_REAL_KEYFRAME
#define MAP_H
#define KEYFRAMEDATABASE_H
#include <algorithm>
#include <climits>
#include <fstream>
#include <iomanip>
#include <limits>
#include <numeric>
#include <sstream>
#include <string>
#include "frame_world.h"
#include "Thirdparty/DBoW2/DBoW2/FORB.h"
#include "Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h"
namespace ORB_SLAM3 {
class MapPoint;
class KeyFrame;
class Map {
public:
    std::mutex mMutexPointCreation, mMutexMapUpdate;
    long unsigned int GetId() { return 0; }
    long unsigned int GetInitKFid() { return 0; }
    bool isImuInitialized() { return false; }
    void EraseMapPoint(MapPoint*) {}
    void EraseKeyFrame(KeyFrame*) {}
#ifdef ORBX_LOCALMAPPING_WORLD     // (localmapping_world.h: what LocalMapping.cc asks of the map, declared only)
    bool GetIniertialBA1(); bool GetIniertialBA2(); void SetIniertialBA1(); void SetIniertialBA2(); void SetImuInitialized();
    void ApplyScaledRotation(const Sophus::SE3f& T, const float s, const bool bScaledVel = false);
    std::vector<KeyFrame*> GetAllKeyFrames(); std::vector<MapPoint*> GetAllMapPoints(); long unsigned int KeyFramesInMap();
    void IncreaseChangeIndex(); void InformNewBigChange(); void AddKeyFrame(KeyFrame*); void AddMapPoint(MapPoint*);
    std::vector<KeyFrame*> mvpKeyFrameOrigins;
    unsigned int GetLowerKFID(); long unsigned int GetMaxKFid(); KeyFrame* GetOriginKF(); void SetCurrentMap(); void SetInertialSensor(); bool IsInertial();
    int GetMapChangeIndex(); int GetLastMapChange(); void SetLastMapChange(int currentChangeId); void SetReferenceMapPoints(const std::vector<MapPoint*>& vpMPs);
    long unsigned int MapPointsInMap();
    void ChangeId(long unsigned int nId); bool CheckEssentialGraph(); void PrintEssentialGraph(); int GetLastBigChangeIdx();
#endif
};
class Frame;
class KeyFrameDatabase {
public:
    void erase(KeyFrame*) {}
#ifdef ORBX_TRACKING_WORLD
    void clear(); void clearMap(Map* pMap); std::vector<KeyFrame*> DetectRelocalizationCandidates(Frame* F, Map* pMap);
    void add(KeyFrame* pKF);
    void DetectNBestCandidates(KeyFrame* pKF, std::vector<KeyFrame*>& vpLoopCand, std::vector<KeyFrame*>& vpMergeCand, int nNumCandidates);
#endif
};
// names used inside the never-instantiated serialize() templates of KeyFrame.h / MapPoint.h
template <class Archive, class T> void serializeMatrix(Archive&, T&, const unsigned int) {}
template <class Archive, class T> void serializeSophusSE3(Archive&, T&, const unsigned int) {}
template <class Archive, class T> void serializeVectorKeyPoints(Archive&, T&, const unsigned int) {}
}
#ifdef ORBX_OPEN_PRIVATE
#define private public
#define protected public
#endif
#endif
