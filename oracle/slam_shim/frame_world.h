// frame_world.h — TEST INFRASTRUCTURE (part of oracle/; never shipped).  Force-included when the reference's own src/Frame.cc + include/Frame.h
// are compiled unmodified and in place (oracle/Makefile, _ref/libref_frame.so): pre-defines the include guards of the headers Frame.h /
// Frame.cc pull in that cannot be compiled here (ImuTypes.h, Converter.h, Settings.h, G2oTypes.h: Eigen, g2o, Boost)
// and supplies just enough of their names for every Frame method to compile.  Only the stereo constructor path is ever executed:
// Frame::Frame(imLeft, imRight, ...) (src/Frame.cc:105-230) -> ExtractORB (the reference's own ORBextractor.cc) -> UndistortKeyPoints
// (no distortion) -> ComputeStereoMatches (:1102-1358) -> AssignFeaturesToGrid, and Frame::GetFeaturesInArea (:859-951).
#ifndef ORBX_FRAME_WORLD_H
#define ORBX_FRAME_WORLD_H

#define IMUTYPES_H
#define CONVERTER_H
#define ORB_SLAM3_SETTINGS_H
#define G2OTYPES_H
#define SERIALIZATION_UTILS_H

#include <iostream>
#include <list>
#include <thread>
#include <opencv2/opencv.hpp>
#include "slam_types.h"

namespace ORB_SLAM3 {
namespace IMU {
#ifdef ORBX_LOCALMAPPING_WORLD     // (localmapping_world.h: the members LocalMapping.cc names, declared only)
struct Bias { float bax = 0, bay = 0, baz = 0, bwx = 0, bwy = 0, bwz = 0; Bias() {} Bias(float, float, float, float, float, float); };
struct Calib { Sophus::SE3f mTcb, mTbc; bool mbIsSet = false; };
struct Preintegrated { void SetNewBias(const Bias&) {} void CopyFrom(Preintegrated*) {} void MergePrevious(Preintegrated*); Eigen::Vector3f GetUpdatedDeltaVelocity(); float dT; };
#else
struct Bias { float bax = 0, bay = 0, baz = 0, bwx = 0, bwy = 0, bwz = 0; };
struct Calib { Sophus::SE3f mTcb, mTbc; bool mbIsSet = false; };
struct Preintegrated { void SetNewBias(const Bias&) {} void CopyFrom(Preintegrated*) {} };
#endif
}
class ConstraintPoseImu {};
class Converter {
public:
    static Eigen::Matrix3f toMatrix3f(const cv::Mat& m) {
        Eigen::Matrix3f r;
        if (!m.empty()) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = m.at<float>(i, j);
        return r;
    }
    static std::vector<cv::Mat> toDescriptorVector(const cv::Mat& D) { std::vector<cv::Mat> v; for (int j = 0; j < D.rows; j++) v.push_back(D.row(j)); return v; }
};
}  // namespace ORB_SLAM3
#endif
