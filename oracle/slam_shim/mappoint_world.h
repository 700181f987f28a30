// mappoint_world.h — TEST INFRASTRUCTURE (part of oracle/; never shipped).  Force-included when the reference's own src/MapPoint.cc +
// include/MapPoint.h are compiled unmodified and in place (oracle/Makefile, _ref/libref_mappoint.so): the real Frame.h is used (over
// frame_world.h), KeyFrame is the stand-in of slam_types.h, Map a three-member stand-in.  Only MapPoint::MapPoint(Pos, pRefKF, pMap),
// AddObservation, ComputeDistinctiveDescriptors (src/MapPoint.cc:438-529) and GetDescriptor are executed.
#ifndef ORBX_MAPPOINT_WORLD_H
#define ORBX_MAPPOINT_WORLD_H
#define ORBX_REAL_MAPPOINT
#define MAP_H
#include <climits>
#include "frame_world.h"
namespace ORB_SLAM3 {
class MapPoint;
class Map {
public:
    std::mutex mMutexPointCreation;
    long unsigned int GetId() { return 0; }
    void EraseMapPoint(MapPoint*) {}
};
}
#endif
