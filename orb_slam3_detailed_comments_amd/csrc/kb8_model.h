// kb8_model.h — the Kannala-Brandt fisheye camera of the reference (src/CameraModels/KannalaBrandt8.cpp) as host/device functions:
// project (:87-104), unproject (:180-216), Triangulate (:553-573) and TriangulateMatches (:439-523), which is also the camera's
// epipolarConstrain (:322-328).  Arithmetic follows the reference statement by statement in fp32 (the few places where the reference mixes
// in double - cos / sin of psi, the comparisons against double literals - do the same here).  The one deviation is the null vector of the
// 4x4 triangulation matrix: the reference takes the last column of V from Eigen::JacobiSVD<Matrix4f>; Eigen is not available, and its
// result is defined only up to fp32 rounding anyway, so the vector is computed here as the eigenvector of A^T A with the smallest
// eigenvalue by a cyclic Jacobi iteration in fp64.  Depths agree with an fp32 SVD to ~1e-6 relative; tests allow 1e-4.
#pragma once
#include <cmath>
#include "orbx_platform.h"
#include "glibc_atan2f_model.h"
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
        scale = tanf(theta) / theta_d;
    }
    r[0] = pwx * scale; r[1] = pwy * scale; r[2] = 1.f;
}

// camera point -> pixel (:87-104).  atan2f = glibc's (glibc_atan2f_model.h: bit for bit, every platform); the double cos / sin of psi are the
// platform's: a last-bit difference in double disappears in the rounding of the products to float (probability ~2^-29 per call)
ORBX_HD inline void kb8_project(const KB8Cam& c, const float p[3], float uv[2]) {
    const float x2_plus_y2 = p[0] * p[0] + p[1] * p[1];
    const float theta = glibc_atan2f_model(sqrtf(x2_plus_y2), p[2]);
    const float psi = glibc_atan2f_model(p[1], p[0]);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
    uv[0] = (float)((double)(c.p[0] * r) * cos((double)psi) + (double)c.p[2]);
    uv[1] = (float)((double)(c.p[1] * r) * sin((double)psi) + (double)c.p[3]);
}

// x with A x = 0 in the least-squares sense (the singular vector of the smallest singular value), A 4x4 row-major
ORBX_HD inline void null_vector4(const float A[16], double x[4]) {
    double M[4][4], V[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += (double)A[4 * k + i] * (double)A[4 * k + j];
            M[i][j] = s; V[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 12; sweep++) {
        double off = 0;
        for (int p = 0; p < 4; p++) for (int q = p + 1; q < 4; q++) off += M[p][q] * M[p][q];
        if (off < 1e-300) break;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                if (M[p][q] == 0.0) continue;
                const double th = (M[q][q] - M[p][p]) / (2.0 * M[p][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 4; k++) { const double a = M[k][p], b = M[k][q]; M[k][p] = cs * a - sn * b; M[k][q] = sn * a + cs * b; }
                for (int k = 0; k < 4; k++) { const double a = M[p][k], b = M[q][k]; M[p][k] = cs * a - sn * b; M[q][k] = sn * a + cs * b; }
                for (int k = 0; k < 4; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = cs * a - sn * b; V[k][q] = sn * a + cs * b; }
            }
    }
    // column of V that belongs to the smallest eigenvalue, by selects: an index computed at run time (V[k][m]) would send the whole of V to
    // private memory (the kernels that inline this carried a 144-byte scratch segment for it)
    double best = M[0][0];
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = V[k][0];
#pragma unroll
    for (int i = 1; i < 4; i++) {
        const bool take = M[i][i] < best;
        best = take ? M[i][i] : best;
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = take ? V[k][i] : x[k];
    }
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
    double xh[4];
    null_vector4(A, xh);
    const float h3 = (float)xh[3];
    const float x3D[3] = {(float)xh[0] / h3, (float)xh[1] / h3, (float)xh[2] / h3};
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
