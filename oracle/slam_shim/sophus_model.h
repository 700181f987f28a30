// sophus_model.h — TEST INFRASTRUCTURE (part of oracle/; never shipped).
//
// Quaternion-backed stand-ins for Sophus::SO3f / SE3f / RxSO3f / Sim3f, restating the reference's VENDORED Sophus
// (/root/reference/Thirdparty/Sophus/sophus/{so3,se3,rxso3,sim3}.hpp) statement by statement, over a stand-in Eigen::Quaternionf that restates
// the Eigen operations those statements call.  The vendored headers themselves cannot be compiled here: they need Eigen (>= 3.3.0,
// Thirdparty/Sophus/CMakeLists.txt:35), which is an external dependency of the reference and is not installed in this image.
//
// What is restated, with the line of the vendored source next to each member:
//   SO3:   default / from-matrix / from-quaternion constructors (so3.hpp:451-488), normalize (:297-303), inverse (:229-231), matrix (:310-312),
//          group product (:325-340, the explicit quaternion formula, result normalised by the quaternion constructor), point action (:357-367)
//   SE3:   constructors (se3.hpp:448-491), inverse (:208-211), product (:304-308), point action (:321-324), rotationMatrix (:363)
//   RxSO3: constructors (rxso3.hpp:451-492), inverse (:156-158), matrix (:191-220), point action (:265-273), rotationMatrix (:341-345), scale (:350)
//   Sim3:  constructors (sim3.hpp:388-413), inverse (:129-132), point action (:226-229), rotationMatrix (:295-297), scale (:313)
// Eigen operations behind them (Eigen 3.3.x / 3.4.0, Eigen/src/Geometry/Quaternion.h and OrthoMethods.h; restated from the published algorithm):
//   Quaternion(Matrix3)        quaternionbase_assign_impl<Other,3,3>::run (Shoemake's branch on the trace)
//   toRotationMatrix()         QuaternionBase::toRotationMatrix (tx = 2x, ... ; 1 - (tyy + tzz), txy - twz, ...)
//   conjugate(), inverse()     conjugate().coeffs() / squaredNorm()
//   squaredNorm()              coeffs().squaredNorm(): a 16-byte aligned Vector4f, i.e. ONE SSE packet on the reference's x86-64 target (Eigen
//                              vectorises by default there); predux<Packet4f> adds the high half onto the low half first:
//                              (x*x + z*z) + (y*y + w*w).  This is the one place where an Eigen-internal summation order enters; the PRODUCT
//                              never computes it (the facade passes the caller's own `scale()` to the device), only this checker does.
//   cross()                    MatrixBase::cross for 3-vectors: (a1 b2 - a2 b1, a2 b0 - a0 b2, a0 b1 - a1 b0)
// Expression templates are evaluated coefficient by coefficient in the order written: `p + q.w() * uv + q.vec().cross(uv)` is
// (p[i] + w * uv[i]) + c[i] with the cross product evaluated into a temporary first (cross() returns a plain object).
// No fused multiply-adds (the oracle is built -ffp-contract=off like the rest of the parity chain).
#ifndef ORBX_SOPHUS_MODEL_H
#define ORBX_SOPHUS_MODEL_H

#include <cmath>

namespace Eigen {

inline Vector3f cross3(const Vector3f& a, const Vector3f& b) {          // Eigen/src/Geometry/OrthoMethods.h, MatrixBase::cross
    return Vector3f(a.d[1] * b.d[2] - a.d[2] * b.d[1], a.d[2] * b.d[0] - a.d[0] * b.d[2], a.d[0] * b.d[1] - a.d[1] * b.d[0]);
}

struct Quaternionf {
#ifdef ORBX_LOOPCLOSING_WORLD      // (loopclosing_world.h: declared only)
    template <class T> struct CastQ;
    template <class T> typename CastQ<T>::type cast() const;
    Vector3f operator*(const Vector3f& v) const;
#endif
    float c[4];                                                          // coeffs(): x, y, z, w
    Quaternionf() : c{0, 0, 0, 1} {}
    Quaternionf(float w, float x, float y, float z) : c{x, y, z, w} {}   // Eigen's constructor takes w first
    explicit Quaternionf(const Matrix3f& mat) {                          // Quaternion.h, quaternionbase_assign_impl<Other,3,3>::run
        float t = mat(0, 0) + (mat(1, 1) + mat(2, 2));                   // trace(): the 3-term reduction is unrolled as a0 + (a1 + a2) (Redux.h)
        if (t > 0.0f) {
            t = std::sqrt(t + 1.0f);
            c[3] = 0.5f * t;
            t = 0.5f / t;
            c[0] = (mat(2, 1) - mat(1, 2)) * t;
            c[1] = (mat(0, 2) - mat(2, 0)) * t;
            c[2] = (mat(1, 0) - mat(0, 1)) * t;
        } else {
            int i = 0;
            if (mat(1, 1) > mat(0, 0)) i = 1;
            if (mat(2, 2) > mat(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(mat(i, i) - mat(j, j) - mat(k, k) + 1.0f);
            c[i] = 0.5f * t;
            t = 0.5f / t;
            c[3] = (mat(k, j) - mat(j, k)) * t;
            c[j] = (mat(j, i) + mat(i, j)) * t;
            c[k] = (mat(k, i) + mat(i, k)) * t;
        }
    }
    float x() const { return c[0]; }
    float y() const { return c[1]; }
    float z() const { return c[2]; }
    float w() const { return c[3]; }
    Vector3f vec() const { return Vector3f(c[0], c[1], c[2]); }
    const float* coeffs() const { return c; }
    float squaredNorm() const { return (c[0] * c[0] + c[2] * c[2]) + (c[1] * c[1] + c[3] * c[3]); }   // one SSE packet, predux<Packet4f> (see header)
    float norm() const { return std::sqrt(squaredNorm()); }
    void normalize() { const float n = norm(); for (float& v : c) v /= n; }                             // coeffs() /= norm()
    void scaleCoeffs(float s) { for (float& v : c) v *= s; }                                             // coeffs() *= s
    Quaternionf conjugate() const { return Quaternionf(c[3], -c[0], -c[1], -c[2]); }
    Quaternionf inverse() const {                                        // Quaternion.h, QuaternionBase::inverse: conjugate().coeffs() / n2
        const float n2 = squaredNorm();
        if (n2 > 0.0f) { const Quaternionf q = conjugate(); return Quaternionf(q.c[3] / n2, q.c[0] / n2, q.c[1] / n2, q.c[2] / n2); }
        return Quaternionf(0, 0, 0, 0);
    }
    Matrix3f toRotationMatrix() const {                                  // Quaternion.h, QuaternionBase::toRotationMatrix
        Matrix3f res;
        const float tx = 2.0f * x(), ty = 2.0f * y(), tz = 2.0f * z();
        const float twx = tx * w(), twy = ty * w(), twz = tz * w();
        const float txx = tx * x(), txy = ty * x(), txz = tz * x();
        const float tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
        res(0, 0) = 1.0f - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
        res(1, 0) = txy + twz; res(1, 1) = 1.0f - (txx + tzz); res(1, 2) = tyz - twx;
        res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = 1.0f - (txx + tyy);
        return res;
    }
};

}  // namespace Eigen

namespace Sophus {

class SO3f {
    Eigen::Quaternionf unit_quaternion_;
public:
#ifdef ORBX_LOCALMAPPING_WORLD     // (localmapping_world.h: declared only)
    static SO3f exp(const Eigen::Vector3f& omega);
    Eigen::Vector3f log() const;
#endif
    SO3f() : unit_quaternion_(1.0f, 0.0f, 0.0f, 0.0f) {}                                           // so3.hpp:451-452
    SO3f(const Eigen::Matrix3f& R) : unit_quaternion_(R) {}                                        // so3.hpp:469 (SOPHUS_ENSUREs only; no normalisation)
    explicit SO3f(const Eigen::Quaternionf& quat) : unit_quaternion_(quat) { normalize(); }        // so3.hpp:480-487
    void normalize() { unit_quaternion_.normalize(); }                                             // so3.hpp:297-303: coeffs() /= norm()
    const Eigen::Quaternionf& unit_quaternion() const { return unit_quaternion_; }
    SO3f inverse() const { return SO3f(unit_quaternion_.conjugate()); }                            // so3.hpp:229-231
    Eigen::Matrix3f matrix() const { return unit_quaternion_.toRotationMatrix(); }                 // so3.hpp:310-312
    static Eigen::Matrix3f hat(const Eigen::Vector3f& omega) {                                     // so3.hpp:673-682
        Eigen::Matrix3f Omega;
        Omega(0, 1) = -omega(2); Omega(0, 2) = omega(1); Omega(1, 0) = omega(2); Omega(1, 2) = -omega(0); Omega(2, 0) = -omega(1); Omega(2, 1) = omega(0);
        return Omega;
    }
    SO3f operator*(const SO3f& other) const {                                                      // so3.hpp:325-340
        const Eigen::Quaternionf& a = unit_quaternion_; const Eigen::Quaternionf& b = other.unit_quaternion_;
        return SO3f(Eigen::Quaternionf(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                                       a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                                       a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                                       a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x()));
    }
    Eigen::Vector3f operator*(const Eigen::Vector3f& p) const {                                    // so3.hpp:357-367
        const Eigen::Quaternionf& q = unit_quaternion_;
        Eigen::Vector3f uv = Eigen::cross3(q.vec(), p);
        for (int i = 0; i < 3; i++) uv.d[i] = uv.d[i] + uv.d[i];                                   // uv += uv
        const Eigen::Vector3f c = Eigen::cross3(q.vec(), uv);
        return Eigen::Vector3f((p.d[0] + q.w() * uv.d[0]) + c.d[0], (p.d[1] + q.w() * uv.d[1]) + c.d[1], (p.d[2] + q.w() * uv.d[2]) + c.d[2]);
    }
};

template <typename T> struct SE3;
template <> struct SE3<float> {
private:
    SO3f so3_; Eigen::Vector3f translation_;
public:
    SE3() : so3_(), translation_() {}                                                              // se3.hpp:448 (identity, zero translation)
    SE3(const SO3f& so3, const Eigen::Vector3f& translation) : so3_(so3), translation_(translation) {}                    // se3.hpp:466-473
    SE3(const Eigen::Matrix3f& rotation_matrix, const Eigen::Vector3f& translation) : so3_(rotation_matrix), translation_(translation) {}   // se3.hpp:480-482
    SE3(const Eigen::Quaternionf& quaternion, const Eigen::Vector3f& translation) : so3_(quaternion), translation_(translation) {}          // se3.hpp:488-490
    const SO3f& so3() const { return so3_; }
    const Eigen::Vector3f& translation() const { return translation_; }
#ifdef ORBX_TRACKING_WORLD         // (tracking_world.h: declared only)
    Eigen::Vector3f& translation();
    template <class M4> explicit SE3(const M4& T);
#endif
#ifdef ORBX_LOOPCLOSING_WORLD
    template <class T> SE3<T> cast() const;
#endif
    const Eigen::Quaternionf& unit_quaternion() const { return so3_.unit_quaternion(); }           // se3.hpp:419-421
    Eigen::Matrix3f rotationMatrix() const { return so3_.matrix(); }                                // se3.hpp:363
    Eigen::Matrix34f matrix3x4() const { Eigen::Matrix34f M; M << rotationMatrix(), translation_; return M; }            // se3.hpp:285-290
    SE3 inverse() const { const SO3f invR = so3_.inverse(); return SE3(invR, invR * (translation_ * -1.0f)); }            // se3.hpp:208-211
    SE3 operator*(const SE3& other) const { return SE3(so3_ * other.so3_, translation_ + so3_ * other.translation_); }     // se3.hpp:304-308
    Eigen::Vector3f operator*(const Eigen::Vector3f& p) const { return so3_ * p + translation_; }                         // se3.hpp:321-324
};
typedef SE3<float> SE3f;
#ifdef ORBX_LOOPCLOSING_WORLD
typedef SE3<double> SE3d;
#endif
#ifdef ORBX_LOOPCLOSING_WORLD      // (loopclosing_world.h: the double-precision poses LoopClosing.cc handles, declared only)
}
namespace Eigen { struct Quaterniond; struct Matrix3d; }
namespace Sophus {
struct SO3d { SO3d(); explicit SO3d(const Eigen::Matrix3d& R); };
template <> struct SE3<double> {
    SE3(); SE3(const Eigen::Quaterniond& q, const Eigen::Vector3d& t); SE3(const Eigen::Matrix3d& R, const Eigen::Vector3d& t);
    SE3 operator*(const SE3& o) const; SE3 inverse() const; const Eigen::Quaterniond& unit_quaternion() const; const Eigen::Vector3d& translation() const; Eigen::Vector3d& translation();
    Eigen::Matrix3d rotationMatrix() const; template <class T> SE3<T> cast() const;
};
#else
template <> struct SE3<double> {};     // include/Frame.h:369 holds an unused Sophus::SE3<double> member
#endif

class RxSO3f {
    Eigen::Quaternionf quaternion_;                                                                // |q|^2 = scale
public:
    RxSO3f() : quaternion_(1.0f, 0.0f, 0.0f, 0.0f) {}                                              // rxso3.hpp:433-434
    RxSO3f(const float& scale, const Eigen::Matrix3f& R) : quaternion_(R) { quaternion_.scaleCoeffs(std::sqrt(scale)); }                 // rxso3.hpp:460-466
    RxSO3f(const float& scale, const SO3f& so3) : quaternion_(so3.unit_quaternion()) { quaternion_.scaleCoeffs(std::sqrt(scale)); }      // rxso3.hpp:472-478
    explicit RxSO3f(const Eigen::Quaternionf& quat) : quaternion_(quat) {}                         // rxso3.hpp:484-491 (SOPHUS_ENSURE only)
    const Eigen::Quaternionf& quaternion() const { return quaternion_; }
    RxSO3f inverse() const { return RxSO3f(quaternion_.inverse()); }                               // rxso3.hpp:156-158
    float scale() const { return quaternion_.squaredNorm(); }                                      // rxso3.hpp:350
    Eigen::Matrix3f rotationMatrix() const { Eigen::Quaternionf n = quaternion_; n.normalize(); return n.toRotationMatrix(); }          // rxso3.hpp:341-345
    Eigen::Matrix3f matrix() const {                                                               // rxso3.hpp:191-220
        Eigen::Matrix3f sR;
        const Eigen::Quaternionf& q = quaternion_;
        const float vx_sq = q.x() * q.x(), vy_sq = q.y() * q.y(), vz_sq = q.z() * q.z(), w_sq = q.w() * q.w();
        const float two_vx = 2.0f * q.x(), two_vy = 2.0f * q.y(), two_vz = 2.0f * q.z();
        const float two_vx_vy = two_vx * q.y(), two_vx_vz = two_vx * q.z(), two_vx_w = two_vx * q.w();
        const float two_vy_vz = two_vy * q.z(), two_vy_w = two_vy * q.w(), two_vz_w = two_vz * q.w();
        sR(0, 0) = vx_sq - vy_sq - vz_sq + w_sq; sR(1, 0) = two_vx_vy + two_vz_w; sR(2, 0) = two_vx_vz - two_vy_w;
        sR(0, 1) = two_vx_vy - two_vz_w; sR(1, 1) = -vx_sq + vy_sq - vz_sq + w_sq; sR(2, 1) = two_vx_w + two_vy_vz;
        sR(0, 2) = two_vx_vz + two_vy_w; sR(1, 2) = -two_vx_w + two_vy_vz; sR(2, 2) = -vx_sq - vy_sq + vz_sq + w_sq;
        return sR;
    }
    Eigen::Vector3f operator*(const Eigen::Vector3f& p) const {                                    // rxso3.hpp:265-273
        const float scale = quaternion_.squaredNorm();
        Eigen::Vector3f two_vec_cross_p = Eigen::cross3(quaternion_.vec(), p);
        for (int i = 0; i < 3; i++) two_vec_cross_p.d[i] = two_vec_cross_p.d[i] + two_vec_cross_p.d[i];
        const Eigen::Vector3f c = Eigen::cross3(quaternion_.vec(), two_vec_cross_p);
        const float w = quaternion_.w();
        return Eigen::Vector3f(scale * p.d[0] + (w * two_vec_cross_p.d[0] + c.d[0]), scale * p.d[1] + (w * two_vec_cross_p.d[1] + c.d[1]),
                               scale * p.d[2] + (w * two_vec_cross_p.d[2] + c.d[2]));
    }
};

template <typename T> struct Sim3;
template <> struct Sim3<float> {
private:
    RxSO3f rxso3_; Eigen::Vector3f translation_;
public:
#ifdef ORBX_LOOPCLOSING_WORLD
    void setScale(const float& scale);
#endif
    Sim3() : rxso3_(), translation_() {}                                                           // sim3.hpp:370
    Sim3(const RxSO3f& rxso3, const Eigen::Vector3f& translation) : rxso3_(rxso3), translation_(translation) {}           // sim3.hpp:388-395
    Sim3(const Eigen::Quaternionf& quaternion, const Eigen::Vector3f& translation) : rxso3_(quaternion), translation_(translation) {}   // sim3.hpp:402-409
    const RxSO3f& rxso3() const { return rxso3_; }
    const Eigen::Quaternionf& quaternion() const { return rxso3_.quaternion(); }                   // sim3.hpp:289-291
    const Eigen::Vector3f& translation() const { return translation_; }
    Eigen::Matrix3f rotationMatrix() const { return rxso3_.rotationMatrix(); }                     // sim3.hpp:295-297
    float scale() const { return rxso3_.scale(); }                                                 // sim3.hpp:313
    Sim3 inverse() const { const RxSO3f invR = rxso3_.inverse(); return Sim3(invR, invR * (translation_ * -1.0f)); }      // sim3.hpp:129-132
    Eigen::Vector3f operator*(const Eigen::Vector3f& p) const { return rxso3_ * p + translation_; }                       // sim3.hpp:226-229
};
typedef Sim3<float> Sim3f;

}  // namespace Sophus
#endif
