// undistort_model.h — cv::undistortPoints as Frame::UndistortKeyPoints / ComputeImageBounds call it (reference src/Frame.cc:1003-1034, :1043-1075:
// CV_32FC2 points, camera matrix K, distortion (k1, k2, p1, p2[, k3]), no rectification, new projection P = K), for host and device.
// OpenCV's calib3d arithmetic restated (OpenCV is not available here: parity unpinned like the other cv:: primitives):
// double precision throughout, ifx = 1 / fx, five fixed-point iterations (the public overload's TermCriteria(MAX_ITER, 5, 0.01)), the
// result projected with K and stored as float.  variant 0: OpenCV >= 3.4.2 (leaves the iteration with the start value when icdist < 0),
// 1: OpenCV 3.2.  Plain IEEE double operations (-ffp-contract=off): host, emulator and GPU give the same bits.
#pragma once
#ifndef ORBX_HD
#define ORBX_HD
#endif

namespace orbx {

struct UndistortParams { double fx, fy, cx, cy, ifx, ify, k[5]; int variant, active; };

ORBX_HD inline void undistort_point(const UndistortParams& U, float uf, float vf, float* ou, float* ov) {
    const double u = uf, v = vf;
    double x = (u - U.cx) * U.ifx, y = (v - U.cy) * U.ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1 + ((U.k[4] * r2 + U.k[1]) * r2 + U.k[0]) * r2);     // numerator (1 + ((k7 r2 + k6) r2 + k5) r2) = 1: no rational terms
        if (U.variant == 0 && icdist < 0) { x = (u - U.cx) * U.ifx; y = (v - U.cy) * U.ify; break; }
        const double deltaX = 2 * U.k[2] * x * y + U.k[3] * (r2 + 2 * x * x);
        const double deltaY = U.k[2] * (r2 + 2 * y * y) + 2 * U.k[3] * x * y;
        x = (x0 - deltaX) * icdist; y = (y0 - deltaY) * icdist;
    }
    // xx = P00 x + P01 y + P02 with P = K (P01 = 0: + 0 * y changes nothing for finite y), ww = 1 / (0 x + 0 y + 1) = 1
    const double xx = U.fx * x + 0.0 * y + U.cx, yy = 0.0 * x + U.fy * y + U.cy;
    *ou = (float)xx; *ov = (float)yy;
}

}  // namespace orbx
