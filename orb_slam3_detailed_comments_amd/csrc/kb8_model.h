// kb8_model.h — the Kannala-Brandt fisheye camera of the reference (src/CameraModels/KannalaBrandt8.cpp) as host/device functions:
// project (:87-104), unproject (:180-216), Triangulate (:553-573) and TriangulateMatches (:439-523), which is also the camera's
// epipolarConstrain (:322-328).  Arithmetic follows the reference statement by statement in fp32, with the library functions it calls modelled
// bit for bit: atan2f (glibc_atan2f_model.h), tanf (glibc_tanf_model.h), and cos / sin of the float psi.  The latter two resolve to the FLOAT
// overloads (cosf / sinf, glibc_sincosf_model.h) in a build of the reference with GCC >= 6: KannalaBrandt8.h:27 includes TwoViewReconstruction.h,
// whose <opencv2/opencv.hpp> reaches <math.h> (opencv2/flann/lsh_table.h), and libstdc++'s <math.h> puts std::cos(float) into the global
// namespace - the same choice the reference file makes when it is compiled into the checker (oracle/_ref, tests/test_kb8.py).
// Triangulate's null vector is the last column of V of Eigen::JacobiSVD<Matrix4f>(A, ComputeFullV): eigen_jacobi_svd4_null below restates
// Eigen 3.3.7's two-sided Jacobi SVD for a square real matrix in fp32 (JacobiSVD.h compute(), RealSvd2x2.h, Jacobi.h makeJacobi /
// apply_rotation_in_the_plane; Eigen is an external dependency of the reference and not available here: restated from the published algorithm),
// so that depths are the reference's to the bit (rounds 1-4 used an fp64 eigen-decomposition and a 1e-4 tolerance).
#pragma once
#include <cmath>
#include "orbx_platform.h"
#include <cfloat>
#include "glibc_atan2f_model.h"
#include "glibc_sincosf_model.h"
#include "glibc_tanf_model.h"
#include "sophus_action.h"      // eig_dot3: Eigen's 3-term reductions are a0 + (a1 + a2)

namespace orbx {

struct KB8Cam { float p[8]; };                  // fx, fy, cx, cy, k0, k1, k2, k3  (KannalaBrandt8::mvParameters)
constexpr float kKB8Precision = 1e-6f;          // KannalaBrandt8::precision (include/CameraModels/KannalaBrandt8.h:45,51,57)

// pixel -> ray with z = 1 (:180-216): Newton iteration on theta + k0 theta^3 + k1 theta^5 + k2 theta^7 + k3 theta^9 = theta_d
ORBX_HD inline void kb8_unproject(const KB8Cam& c, float u, float v, float r[3]) {
    const float pwx = (u - c.p[2]) / c.p[0], pwy = (v - c.p[3]) / c.p[1];
    float scale = 1.f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    theta_d = fminf(fmaxf((float)(-3.1415926535897932384626433832795 / 2.f), theta_d), (float)(3.1415926535897932384626433832795 / 2.f));
    if ((double)theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const float k0_theta2 = c.p[4] * theta2, k1_theta4 = c.p[5] * theta4, k2_theta6 = c.p[6] * theta6, k3_theta8 = c.p[7] * theta8;
            const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                    (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < kKB8Precision) break;
        }
        scale = glibc_tanf_model(theta) / theta_d;
    }
    r[0] = pwx * scale; r[1] = pwy * scale; r[2] = 1.f;
}

// camera point -> pixel (:87-104).  atan2f, cosf, sinf = glibc's (bit-for-bit models); everything in float, as the expression
// mvParameters[0] * r * cos(psi) + mvParameters[2] is when cos(float) is the float overload
ORBX_HD inline void kb8_project(const KB8Cam& c, const float p[3], float uv[2]) {
    const float x2_plus_y2 = p[0] * p[0] + p[1] * p[1];
    const float theta = glibc_atan2f_model(sqrtf(x2_plus_y2), p[2]);
    const float psi = glibc_atan2f_model(p[1], p[0]);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
    uv[0] = c.p[0] * r * glibc_cosf(psi) + c.p[2];
    uv[1] = c.p[1] * r * glibc_sinf(psi) + c.p[3];
}

// Last column of V of Eigen::JacobiSVD<Matrix4f>(A, ComputeFullV), A 4x4 row-major: the right singular vector of the smallest singular value.
// Eigen 3.3.7, all in float: W = A / max|A|; sweeps over (p, q), p = 1..3, q = 0..p-1, while any |W(p,q)|, |W(q,p)| exceeds
// max(FLT_MIN, 2 eps * maxDiagEntry); each 2x2 block [W(p,p) W(p,q); W(q,p) W(q,q)] is first made symmetric by a rotation from the left
// (RealSvd2x2.h), then diagonalised by makeJacobi; the left rotation acts on rows p, q of W, the right one on columns p, q of W and of V;
// finally the singular values |W(i,i)| are sorted in descending order, the columns of V moving along.
ORBX_HD inline void eigen_rot_apply(float& x, float& y, float c, float s) { const float xi = x, yi = y; x = c * xi + s * yi; y = -s * xi + c * yi; }
ORBX_HD inline void eigen_jacobi_svd4_null(const float A[16], float x[4]) {
    float W[4][4], V[4][4];
    float scale = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) scale = fmaxf(scale, fabsf(A[i]));
    if (scale == 0.f) scale = 1.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { W[i][j] = A[4 * i + j] / scale; V[i][j] = i == j ? 1.f : 0.f; }
    float maxDiagEntry = fmaxf(fmaxf(fabsf(W[0][0]), fabsf(W[1][1])), fmaxf(fabsf(W[2][2]), fabsf(W[3][3])));
    const float precision = 2.f * FLT_EPSILON, considerAsZero = FLT_MIN;
    // Eigen sweeps until nothing is left above the threshold, without a limit; so does this model - up to 64 sweeps (a convergent input takes 3 to 6, so
    // the cap never changes a result; it keeps a workgroup of a device kernel from spinning forever on an input whose fp32 rotations stop making
    // progress: the pair is then triangulated from the current V like Eigen's would be if it were stopped there, and fails the reference's own
    // parallax / depth / reprojection gates or not on its merits)
    bool finished = false;
    for (int sweep = 0; sweep < 64 && !finished; sweep++) {
        finished = true;
#pragma unroll
        for (int p = 1; p < 4; p++)
#pragma unroll
            for (int q = 0; q < p; q++) {
                const float threshold = fmaxf(considerAsZero, precision * maxDiagEntry);
                if (fabsf(W[p][q]) > threshold || fabsf(W[q][p]) > threshold) {
                    finished = false;
                    // real_2x2_jacobi_svd: rot1 makes the block symmetric ...
                    float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
                    const float t = m00 + m11, d = m10 - m01;
                    float c1, s1;
                    if (fabsf(d) < FLT_MIN) { s1 = 0.f; c1 = 1.f; }
                    else { const float u = t / d; const float tmp = sqrtf(1.f + u * u); s1 = 1.f / tmp; c1 = u / tmp; }
                    if (!(c1 == 1.f && s1 == 0.f)) { eigen_rot_apply(m00, m10, c1, s1); eigen_rot_apply(m01, m11, c1, s1); }
                    // ... j_right = makeJacobi(m00, m01, m11) diagonalises it
                    float cr, sr;
                    const float deno = 2.f * fabsf(m01);
                    if (deno < FLT_MIN) { cr = 1.f; sr = 0.f; }
                    else {
                        const float tau = (m00 - m11) / deno;
                        const float w = sqrtf(tau * tau + 1.f);
                        const float tt = tau > 0.f ? 1.f / (tau + w) : 1.f / (tau - w);
                        const float sign_t = tt > 0.f ? 1.f : -1.f;
                        const float n = 1.f / sqrtf(tt * tt + 1.f);
                        sr = -sign_t * (m01 / fabsf(m01)) * fabsf(tt) * n;
                        cr = n;
                    }
                    // j_left = rot1 * j_right.transpose()
                    const float cl = c1 * cr - s1 * (-sr), sl = c1 * (-sr) + s1 * cr;
                    if (!(cl == 1.f && sl == 0.f)) {
#pragma unroll
                        for (int k = 0; k < 4; k++) eigen_rot_apply(W[p][k], W[q][k], cl, sl);             // applyOnTheLeft(p, q, j_left)
                    }
                    if (!(cr == 1.f && -sr == 0.f)) {                                                       // applyOnTheRight(p, q, j_right): j_right.transpose() on columns
#pragma unroll
                        for (int k = 0; k < 4; k++) eigen_rot_apply(W[k][p], W[k][q], cr, -sr);
#pragma unroll
                        for (int k = 0; k < 4; k++) eigen_rot_apply(V[k][p], V[k][q], cr, -sr);
                    }
                    maxDiagEntry = fmaxf(maxDiagEntry, fmaxf(fabsf(W[p][p]), fabsf(W[q][q])));
                }
            }
    }
    float sv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) sv[i] = fabsf(W[i][i]) * scale;
    // selection sort, descending, first maximum wins; static indices only (a run-time column index would send V to private memory)
#pragma unroll
    for (int i = 0; i < 3; i++) {
        int pos = i; float best = sv[i];
#pragma unroll
        for (int k = i + 1; k < 4; k++) if (sv[k] > best) { best = sv[k]; pos = k; }
#pragma unroll
        for (int k = i + 1; k < 4; k++)
            if (pos == k) {
                const float ts = sv[i]; sv[i] = sv[k]; sv[k] = ts;
#pragma unroll
                for (int r = 0; r < 4; r++) { const float tv = V[r][i]; V[r][i] = V[r][k]; V[r][k] = tv; }
            }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) x[r] = V[r][3];
}

// KannalaBrandt8::TriangulateMatches (:439-523) with the two rays already unprojected (r1 by this camera, r2 by camera 2).
// R12 row-major.  Returns the depth in camera 1 (> 0) or the reference's negative rejection codes; p3D is written on success.
ORBX_HD inline float kb8_triangulate_matches(const KB8Cam& c1, const KB8Cam& c2, const float r1[3], const float r2[3], float u1, float v1, float u2, float v2,
                                             const float R12[9], const float t12[3], float sigmaLevel, float unc, float p3D[3]) {
    const float r21[3] = {eig_dot3(R12[0], R12[1], R12[2], r2[0], r2[1], r2[2]), eig_dot3(R12[3], R12[4], R12[5], r2[0], r2[1], r2[2]),
                          eig_dot3(R12[6], R12[7], R12[8], r2[0], r2[1], r2[2])};
    const float n1 = sqrtf(eig_dot3(r1[0], r1[1], r1[2], r1[0], r1[1], r1[2])), n21 = sqrtf(eig_dot3(r21[0], r21[1], r21[2], r21[0], r21[1], r21[2]));
    const float cosParallaxRays = eig_dot3(r1[0], r1[1], r1[2], r21[0], r21[1], r21[2]) / (n1 * n21);
    if ((double)cosParallaxRays > 0.9998) return -1;
    // Tcw1 = [I | 0], Tcw2 = [R21 | -R21 t12]
    float R21[9], tc[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R21[3 * i + j] = R12[3 * j + i];
    for (int i = 0; i < 3; i++) tc[i] = -eig_dot3(R21[3 * i], R21[3 * i + 1], R21[3 * i + 2], t12[0], t12[1], t12[2]);
    // Triangulate (:553-573): A.row(0) = p1.x * Tcw1.row(2) - Tcw1.row(0), ... with p1 = r1.xy, p2 = r2.xy
    float A[16];
    A[0] = r1[0] * 0.f - 1.f; A[1] = r1[0] * 0.f - 0.f; A[2] = r1[0] * 1.f - 0.f; A[3] = r1[0] * 0.f - 0.f;
    A[4] = r1[1] * 0.f - 0.f; A[5] = r1[1] * 0.f - 1.f; A[6] = r1[1] * 1.f - 0.f; A[7] = r1[1] * 0.f - 0.f;
    for (int j = 0; j < 3; j++) { A[8 + j] = r2[0] * R21[6 + j] - R21[j]; A[12 + j] = r2[1] * R21[6 + j] - R21[3 + j]; }
    A[11] = r2[0] * tc[2] - tc[0]; A[15] = r2[1] * tc[2] - tc[1];
    float xh[4];
    eigen_jacobi_svd4_null(A, xh);
    const float x3D[3] = {xh[0] / xh[3], xh[1] / xh[3], xh[2] / xh[3]};
    const float z1 = x3D[2];
    if (!(z1 > 0)) return -2;
    const float z2 = eig_dot3(R21[6], R21[7], R21[8], x3D[0], x3D[1], x3D[2]) + tc[2];
    if (!(z2 > 0)) return -3;
    float uv1[2];
    kb8_project(c1, x3D, uv1);
    const float errX1 = uv1[0] - u1, errY1 = uv1[1] - v1;
    if ((double)(errX1 * errX1 + errY1 * errY1) > 5.991 * (double)sigmaLevel) return -4;
    float x3D2[3];
    eig_rt3(R21, tc, x3D[0], x3D[1], x3D[2], x3D2);
    float uv2[2];
    kb8_project(c2, x3D2, uv2);
    const float errX2 = uv2[0] - u2, errY2 = uv2[1] - v2;
    if ((double)(errX2 * errX2 + errY2 * errY2) > 5.991 * (double)unc) return -5;
    p3D[0] = x3D[0]; p3D[1] = x3D[1]; p3D[2] = x3D[2];
    return z1;
}

}  // namespace orbx
