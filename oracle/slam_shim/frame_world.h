// frame_world.h — TEST INFRASTRUCTURE (part of oracle/; never shipped).  Force-included when the reference's own src/Frame.cc + include/Frame.h
// are compiled unmodified and in place (oracle/Makefile, _ref/libref_frame.so): pre-defines the include guards of the headers Frame.h /
// Frame.cc pull in that cannot be compiled here (ImuTypes.h, Converter.h, Settings.h, G2oTypes.h, the camera models: Eigen, g2o, Boost)
// and supplies just enough of their names for every Frame method to compile.  Only the stereo constructor path is ever executed:
// Frame::Frame(imLeft, imRight, ...) (src/Frame.cc:105-230) -> ExtractORB (the reference's own ORBextractor.cc) -> UndistortKeyPoints
// (no distortion) -> ComputeStereoMatches (:1102-1358) -> AssignFeaturesToGrid, and Frame::GetFeaturesInArea (:859-951).
#ifndef ORBX_FRAME_WORLD_H
#define ORBX_FRAME_WORLD_H

#define IMUTYPES_H
#define CONVERTER_H
#define ORB_SLAM3_SETTINGS_H
#define G2OTYPES_H
#define CAMERAMODELS_GEOMETRICCAMERA_H
#define CAMERAMODELS_PINHOLE_H
#define CAMERAMODELS_KANNALABRANDT8_H
#define SERIALIZATION_UTILS_H

#include <iostream>
#include <list>
#include <thread>
#include <opencv2/opencv.hpp>
#include "slam_types.h"

namespace Eigen {
template <typename T, int R, int C> struct MatrixSel;
template <> struct MatrixSel<float, 3, 1> { typedef Vector3f type; };
template <> struct MatrixSel<float, 3, 3> { typedef Matrix3f type; };
template <> struct MatrixSel<float, 2, 1> { typedef Vector2f type; };
template <> struct MatrixSel<float, 1, 3> { typedef Vector3f type; };
template <typename T, int R, int C> using Matrix = typename MatrixSel<T, R, C>::type;
}

namespace ORB_SLAM3 {
namespace IMU {
struct Bias { float bax = 0, bay = 0, baz = 0, bwx = 0, bwy = 0, bwz = 0; };
struct Calib { Sophus::SE3f mTcb, mTbc; bool mbIsSet = false; };
struct Preintegrated { void SetNewBias(const Bias&) {} void CopyFrom(Preintegrated*) {} };
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
class Pinhole : public GeometricCamera {
public:
    Pinhole(float fx, float fy, float cx, float cy) : GeometricCamera(fx, fy, cx, cy) {}
    cv::Mat toK() { cv::Mat K(3, 3, CV_32F); for (int i = 0; i < 9; i++) K.at<float>(i / 3, i % 3) = 0; K.at<float>(0, 0) = mvParameters[0]; K.at<float>(1, 1) = mvParameters[1];
                    K.at<float>(0, 2) = mvParameters[2]; K.at<float>(1, 2) = mvParameters[3]; K.at<float>(2, 2) = 1; return K; }
};
// KannalaBrandt8: the restated camera of kb8_camera.h (slam_types.h), triangulation gate included
}  // namespace ORB_SLAM3
#endif
