// tests/cpp/eigen_probe.cpp — TEST INFRASTRUCTURE.  One source, two builds (tests/test_real_eigen.py):
//   -DPROBE_REAL_EIGEN : against a real Eigen (>= 3.3) and the reference's vendored Sophus headers,
//   otherwise          : against the stand-ins of oracle/slam_shim (slam_types.h, eigen_small.h, sophus_model.h).
// Both print the bit patterns of the same seeded computations - the operations the reference's matchers, Frame and camera models evaluate
// through Eigen / Sophus: 3-term reductions (dot, norm, Matrix3f * Vector3f), the 3x3 inverse and products of Pinhole::epipolarConstrain, the
// SE3 point action / product / inverse on unit quaternions, rotationMatrix(), and the null vector of JacobiSVD<Matrix4f> (KannalaBrandt8::Triangulate).
// Where a real Eigen exists the two outputs must be identical: that is the independent check of the restated evaluation orders.
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cmath>
#ifdef PROBE_REAL_EIGEN
#include <Eigen/Dense>
#include <sophus/se3.hpp>
#else
#include <vector>
#include <opencv2/opencv.hpp>
#include "slam_types.h"
#endif

static uint32_t rng_state = 12345u;
static float rnd() { rng_state = rng_state * 1664525u + 1013904223u; return (float)((rng_state >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
static void put(const char* tag, float v) { uint32_t u; memcpy(&u, &v, 4); printf("%s %08x\n", tag, u); }

int main() {
    for (int it = 0; it < 2000; it++) {
        Eigen::Matrix3f A, B; Eigen::Vector3f v, w, t;
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) { A(i, j) = rnd() * 3.f; B(i, j) = rnd() * 2.f; } v(i) = rnd() * 10.f; w(i) = rnd() * 10.f; t(i) = rnd(); }
        A(0, 0) += 2.f; A(1, 1) += 2.f; A(2, 2) += 2.f;
        put("dot", v.dot(w)); put("norm", v.norm());
        Eigen::Vector3f av = A * v; for (int i = 0; i < 3; i++) put("Av", av(i));
        Eigen::Vector3f avt = A * v + t; for (int i = 0; i < 3; i++) put("Av+t", avt(i));
        Eigen::Matrix3f inv = A.inverse(); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) put("inv", inv(i, j));
        Eigen::Matrix3f F = A.transpose().inverse() * B * A * B.inverse(); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) put("F", F(i, j));
        // a rotation from three angles, entered through Sophus' SE3(R, t) constructor
        const float a = rnd() * 6.f, b = rnd() * 3.f, c = rnd() * 6.f;
        const float ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b), cc = std::cos(c), sc = std::sin(c);
        Eigen::Matrix3f R;
        R(0, 0) = ca * cb; R(0, 1) = ca * sb * sc - sa * cc; R(0, 2) = ca * sb * cc + sa * sc;
        R(1, 0) = sa * cb; R(1, 1) = sa * sb * sc + ca * cc; R(1, 2) = sa * sb * cc - ca * sc;
        R(2, 0) = -sb;     R(2, 1) = cb * sc;                R(2, 2) = cb * cc;
        Sophus::SE3f T(R, t), T2(R.transpose(), w * 0.1f);
        Eigen::Vector3f p = T * v; for (int i = 0; i < 3; i++) put("T*p", p(i));
        Eigen::Vector3f q = (T * T2).inverse() * v; for (int i = 0; i < 3; i++) put("(TT2)^-1*p", q(i));
        Eigen::Matrix3f Rq = T.rotationMatrix(); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) put("R(q)", Rq(i, j));
        Eigen::Vector3f ti = T.inverse().translation(); for (int i = 0; i < 3; i++) put("Tinv.t", ti(i));
        // the DLT matrix of KannalaBrandt8::Triangulate for a random pair of normalised points, and its null vector
        Eigen::Matrix4f M;
        const float x1 = rnd(), y1 = rnd(), x2 = x1 - 0.05f * (rnd() + 0.6f), y2 = y1 + 0.01f * rnd();
        for (int j = 0; j < 4; j++) {
            const float r0 = j < 3 ? Rq(0, j) : t(0) * 0.2f, r1 = j < 3 ? Rq(1, j) : t(1) * 0.2f, r2 = j < 3 ? Rq(2, j) : t(2) * 0.2f;
            M(0, j) = x1 * (j == 2 ? 1.f : 0.f) - (j == 0 ? 1.f : 0.f);
            M(1, j) = y1 * (j == 2 ? 1.f : 0.f) - (j == 1 ? 1.f : 0.f);
            M(2, j) = x2 * r2 - r0;
            M(3, j) = y2 * r2 - r1;
        }
        Eigen::JacobiSVD<Eigen::Matrix4f> svd(M, Eigen::ComputeFullV);
        Eigen::Vector4f nv = svd.matrixV().col(3); for (int i = 0; i < 4; i++) put("V3", nv(i));
    }
    return 0;
}
