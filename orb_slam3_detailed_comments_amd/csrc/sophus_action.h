// sophus_action.h — the group actions of the reference's vendored Sophus on 3-points, and Eigen's fixed-size 3-term reductions, as host/device
// functions in the reference's fp32 operation order (the library is built -ffp-contract=off: no fused multiply-adds).
//
// The reference evaluates `Tcw * p3Dw` with Sophus types (src/ORBmatcher.cc:533, :652, :1379, :1583, :1757, :1844, :1970, :1987, :2093, :2224).
// Sophus stores a rotation as a unit quaternion and rotates a point WITHOUT forming the matrix
// (Thirdparty/Sophus/sophus/so3.hpp:357-367):
//     uv = q.vec().cross(p);  uv += uv;  return p + q.w() * uv + q.vec().cross(uv);
// SE3 adds the translation (se3.hpp:321-324); RxSO3 - the rotation part of a Sim3 - scales by |q|^2 (rxso3.hpp:265-273):
//     scale = |q|^2;  tv = q.vec().cross(p);  tv += tv;  return scale * p + (q.w() * tv + q.vec().cross(tv));
// and Sim3 adds its translation (sim3.hpp:226-229).  The rounding of these forms differs from `R * p + t` with R = q.toRotationMatrix() in the
// last bit, and a last bit of u or v moves a keypoint in or out of GetFeaturesInArea's window - so the device evaluates exactly these statements.
// Eigen's cross product of 3-vectors is (a1 b2 - a2 b1, a2 b0 - a0 b2, a0 b1 - a1 b0) (Eigen/src/Geometry/OrthoMethods.h); an expression like
// p + w * uv + c is evaluated coefficient by coefficient, left to right: (p_i + w * uv_i) + c_i.
//
// Where the reference itself works with matrices (Frame::isInFrustum: mRcw * P + mtcw, src/Frame.cc:685; KeyFrame / Pinhole algebra), the
// device does too; those 3-term sums - and every dot(), norm() and trace() of 3-vectors - are Eigen reductions.  The vendored Sophus requires
// Eigen >= 3.3.0 (Thirdparty/Sophus/CMakeLists.txt:35); from that version on a fixed-size product coefficient is
// `(lhs.row(i).transpose().cwiseProduct(rhs.col(j))).sum()` (Eigen/src/Core/ProductEvaluators.h) and a fixed-size sum() of three terms is unrolled
// by redux_novec_unroller, which splits the range in halves (Eigen/src/Core/Redux.h): a0 + (a1 + a2), not (a0 + a1) + a2.
#pragma once
#include "orbx_platform.h"

namespace orbx {

ORBX_HD inline float eig_sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }
ORBX_HD inline float eig_dot3(float a0, float a1, float a2, float b0, float b1, float b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }
// row-major 3x3 times vector plus translation: the expression `R * p + t` (the product is evaluated first, then the sum, per coefficient)
ORBX_HD inline void eig_rt3(const float R[9], const float t[3], float p0, float p1, float p2, float out[3]) {
    out[0] = eig_dot3(R[0], R[1], R[2], p0, p1, p2) + t[0];
    out[1] = eig_dot3(R[3], R[4], R[5], p0, p1, p2) + t[1];
    out[2] = eig_dot3(R[6], R[7], R[8], p0, p1, p2) + t[2];
}

// q = coeffs() order (x, y, z, w)
// Sophus::SO3::operator*(point), so3.hpp:357-367
ORBX_HD inline void so3_act(const float q[4], float p0, float p1, float p2, float out[3]) {
    float u0 = q[1] * p2 - q[2] * p1, u1 = q[2] * p0 - q[0] * p2, u2 = q[0] * p1 - q[1] * p0;      // uv = q.vec().cross(p)
    u0 = u0 + u0; u1 = u1 + u1; u2 = u2 + u2;                                                       // uv += uv
    const float c0 = q[1] * u2 - q[2] * u1, c1 = q[2] * u0 - q[0] * u2, c2 = q[0] * u1 - q[1] * u0; // q.vec().cross(uv)
    out[0] = (p0 + q[3] * u0) + c0; out[1] = (p1 + q[3] * u1) + c1; out[2] = (p2 + q[3] * u2) + c2;
}
// Sophus::SE3::operator*(point), se3.hpp:321-324: so3() * p + translation()
ORBX_HD inline void se3_act(const float q[4], const float t[3], float p0, float p1, float p2, float out[3]) {
    so3_act(q, p0, p1, p2, out);
    out[0] = out[0] + t[0]; out[1] = out[1] + t[1]; out[2] = out[2] + t[2];
}
// Sophus::Sim3::operator*(point), sim3.hpp:226-229 over RxSO3::operator*(point), rxso3.hpp:265-273.  `scale` = quaternion().squaredNorm() is the
// caller's own Sim3::scale() (the same expression, evaluated by the caller's Eigen), so no Eigen-internal summation order is assumed here.
ORBX_HD inline void sim3_act(const float q[4], float scale, const float t[3], float p0, float p1, float p2, float out[3]) {
    float u0 = q[1] * p2 - q[2] * p1, u1 = q[2] * p0 - q[0] * p2, u2 = q[0] * p1 - q[1] * p0;
    u0 = u0 + u0; u1 = u1 + u1; u2 = u2 + u2;
    const float c0 = q[1] * u2 - q[2] * u1, c1 = q[2] * u0 - q[0] * u2, c2 = q[0] * u1 - q[1] * u0;
    out[0] = (scale * p0 + (q[3] * u0 + c0)) + t[0]; out[1] = (scale * p1 + (q[3] * u1 + c1)) + t[1]; out[2] = (scale * p2 + (q[3] * u2 + c2)) + t[2];
}

// TEST SWITCH (orbx_debug_stereo_flags bit 4): round 3's form of the same transform - R = q.toRotationMatrix() (Eigen/src/Geometry/Quaternion.h),
// then R * p + t with the three products summed left to right.  Mathematically the same point; its last bits differ, and
// tests/test_sophus_action.py requires the reference to CATCH it on map points placed on search-window edges.
ORBX_HD inline void se3_act_matrix_form(const float q[4], const float t[3], float p0, float p1, float p2, float out[3]) {
    const float tx = 2.0f * q[0], ty = 2.0f * q[1], tz = 2.0f * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3], txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    const float R[9] = {1.0f - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0f - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1.0f - (txx + tyy)};
    out[0] = ((R[0] * p0 + R[1] * p1) + R[2] * p2) + t[0]; out[1] = ((R[3] * p0 + R[4] * p1) + R[5] * p2) + t[1]; out[2] = ((R[6] * p0 + R[7] * p1) + R[8] * p2) + t[2];
}

}  // namespace orbx
