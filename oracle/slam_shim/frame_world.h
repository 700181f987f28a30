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
#ifdef ORBX_LOCALMAPPING_WORLD     // (localmapping_world.h / tracking_world.h: the members LocalMapping.cc and Tracking.cc name, declared only)
struct Bias { float bax = 0, bay = 0, baz = 0, bwx = 0, bwy = 0, bwz = 0; Bias() {} Bias(float, float, float, float, float, float); };
struct Calib { Sophus::SE3f mTcb, mTbc; bool mbIsSet = false; Calib() {} Calib(const Sophus::SE3f& Tbc, const float& ng, const float& na, const float& ngw, const float& naw); };
struct Point { Point(const float&, const float&, const float&, const float&, const float&, const float&, const double&); Point(const cv::Point3f, const cv::Point3f, const double&);
               Eigen::Vector3f a, w; double t; };
const float GRAVITY_VALUE = 9.81;
Eigen::Matrix3f NormalizeRotation(const Eigen::Matrix3f& R);
struct Preintegrated {
    Preintegrated() {} Preintegrated(const Bias& b_, const Calib& calib); Preintegrated(Preintegrated* pImuPre);
    void SetNewBias(const Bias&) {} void CopyFrom(Preintegrated*) {} void MergePrevious(Preintegrated*);
    void IntegrateNewMeasurement(const Eigen::Vector3f& acceleration, const Eigen::Vector3f& angVel, const float& dt);
    Eigen::Vector3f GetUpdatedDeltaVelocity(); Eigen::Matrix3f GetUpdatedDeltaRotation(); Eigen::Vector3f GetUpdatedDeltaPosition();
    Eigen::Vector3f GetDeltaVelocity(const Bias& b_); Eigen::Matrix3f GetDeltaRotation(const Bias& b_); Eigen::Vector3f GetDeltaPosition(const Bias& b_);
    Bias GetUpdatedBias();
    float dT; Eigen::Vector3f avgA, avgW;
};
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
#ifdef ORBX_TRACKING_WORLD
    static Sophus::SE3<float> toSophus(const cv::Mat& T);
#endif
#ifdef ORBX_LOOPCLOSING_WORLD
    template <class S3> static Sophus::Sim3<float> toSophus(const S3& S);          // (g2o::Sim3: declared in loopclosing_world.h)
#endif
};
}  // namespace ORB_SLAM3
#endif
