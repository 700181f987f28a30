// oracle/orb_primitives.h — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// Plain scalar C++ restatement of the OpenCV primitives that the reference ORB front-end calls.
// OpenCV is an un-vendored external dependency of the reference (CMakeLists.txt:35
// `find_package(OpenCV 3.2)`, README.md:72 "Tested with OpenCV 3.2.0 and 4.4.0") and is NOT
// installed in this image, so the arithmetic below is restated from OpenCV's published scalar
// algorithms (imgproc/resize.cpp, features2d/fast.cpp + fast_score.cpp, imgproc/smooth, core
// mathfuncs fastAtan2).  PARITY UNPINNED against a real OpenCV build: there are no golden vectors
// in the reference for this path.  Each primitive is cross-checked against an
// independent slow definition in tests/test_oracle_primitives.py.
//
// Call sites in the reference that these stand in for:
//   cv::resize(INTER_LINEAR)            src/ORBextractor.cc:1702
//   cv::copyMakeBorder(REFLECT_101)     src/ORBextractor.cc:1712, :1734
//   cv::FAST(img,kps,th,true)           src/ORBextractor.cc:1135, :1144
//   cv::GaussianBlur(7x7, s=2, R101)    src/ORBextractor.cc:1632
//   cv::fastAtan2                       src/ORBextractor.cc:137
//   cvRound/cvFloor/cvCeil              src/ORBextractor.cc:97,160,168,519,547,553,559,1692
//   cv::norm(NORM_L1)                   src/Frame.cc:1278
#pragma once
#include <algorithm>
#include <cfloat>
#include <cstddef>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace orbp {
using std::ptrdiff_t;

// cvRound: round-half-to-even (SSE cvtsd2si semantics under the default rounding mode).
static inline int round_half_even(double v) { return (int)lrint(v); }
static inline int floor_i(double v) { int i = (int)v; return i - (i > v); }
static inline int ceil_i(double v) { int i = (int)v; return i + (i < v); }
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

// ---- cv::resize, INTER_LINEAR, 8UC1 (fixed point, 11-bit coefficients) -----------------------
struct ResizeTables {
    std::vector<int> xofs, yofs;         // source column / row index
    std::vector<short> ialpha, ibeta;    // 2 coefficients per destination column / row
};
static inline void resize_tables(int sw, int sh, int dw, int dh, ResizeTables& t) {
    const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    const double scale_x = 1.0 / inv_sx, scale_y = 1.0 / inv_sy;
    t.xofs.resize(dw); t.yofs.resize(dh); t.ialpha.resize(2 * dw); t.ibeta.resize(2 * dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = floor_i(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        t.xofs[dx] = sx;
        t.ialpha[2 * dx] = sat_short(round_half_even((1.f - fx) * 2048.f));
        t.ialpha[2 * dx + 1] = sat_short(round_half_even(fx * 2048.f));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = floor_i(fy);
        fy -= sy;
        t.yofs[dy] = sy;   // rows are clamped at use; the weights are NOT modified (OpenCV behaviour)
        t.ibeta[2 * dy] = sat_short(round_half_even((1.f - fy) * 2048.f));
        t.ibeta[2 * dy + 1] = sat_short(round_half_even(fy * 2048.f));
    }
}
static inline void resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstep,
                                    uint8_t* dst, int dw, int dh, size_t dstep) {
    ResizeTables t; resize_tables(sw, sh, dw, dh, t);
    std::vector<int> rows[2]; rows[0].resize(dw); rows[1].resize(dw);
    for (int dy = 0; dy < dh; dy++) {
        for (int k = 0; k < 2; k++) {
            int sy = t.yofs[dy] + k;
            sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
            const uint8_t* S = src + (size_t)sy * sstep;
            for (int dx = 0; dx < dw; dx++) {
                int sx = t.xofs[dx];
                int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;   // a1==0 whenever sx==sw-1
                rows[k][dx] = S[sx] * t.ialpha[2 * dx] + S[sx1] * t.ialpha[2 * dx + 1];
            }
        }
        const int b0 = t.ibeta[2 * dy], b1 = t.ibeta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstep;
        for (int dx = 0; dx < dw; dx++) {
            int v = (((b0 * (rows[0][dx] >> 4)) >> 16) + ((b1 * (rows[1][dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

// ---- cv::remap(src, dst, map1 (CV_32FC1 x), map2 (CV_32FC1 y), INTER_LINEAR, BORDER_CONSTANT, Scalar()) for 8U, cn channels ----
// OpenCV imgproc/remap: float maps are converted block-wise to fixed point, sx = cvRound(x * INTER_TAB_SIZE) with
// INTER_TAB_SIZE = 32, integer part saturate_cast<short>(sx >> 5), fraction sx & 31; weights from BilinearTab_i
// (saturate_cast<short>((1-fy)(1-fx) * 32768) ..., exact integers for 5-bit fractions; the single table entry whose sum
// misses 32768 - fx = fy = 0, weight 32767 - gets the missing unit on tap 3, which cannot change an 8-bit result);
// result = saturate_cast<uchar>((sum + (1 << 14)) >> 15); taps outside the source read the border value 0.
static inline void remap_linear_u8(const uint8_t* src, int sw, int sh, size_t sstep, int cn, const float* mapx, const float* mapy,
                                   uint8_t* dst, int dw, int dh, size_t dstep) {
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            const int fsx = round_half_even((double)(mapx[(size_t)y * dw + x] * 32.0f)), fsy = round_half_even((double)(mapy[(size_t)y * dw + x] * 32.0f));
            const int sx = sat_short(fsx >> 5), sy = sat_short(fsy >> 5);
            const int fx = fsx & 31, fy = fsy & 31;
            int w[4] = {(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32};
            if (fx == 0 && fy == 0) { w[0] = 32767; w[3] = 1; }
            for (int c = 0; c < cn; c++) {
                int v[4];
                for (int k = 0; k < 4; k++) {
                    const int xx = sx + (k & 1), yy = sy + (k >> 1);
                    v[k] = (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[(size_t)yy * sstep + (size_t)xx * cn + c] : 0;
                }
                const int r = (v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3] + (1 << 14)) >> 15;
                dst[(size_t)y * dstep + (size_t)x * cn + c] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
        }
}

// ---- cv::cvtColor(src, dst, COLOR_RGB2GRAY / BGR2GRAY / RGBA2GRAY / BGRA2GRAY) for 8U ----
// variant 0: OpenCV 4.x (RY15 = 9798, GY15 = 19235, BY15 = 3735, gray_shift = 15); variant 1: OpenCV 3.x (R2Y = 4899, G2Y = 9617,
// B2Y = 1868, yuv_shift = 14); dst = CV_DESCALE(r*RY + g*GY + b*BY, shift).
static inline void cvt_gray_u8(const uint8_t* src, int w, int h, size_t sstep, int cn, int red_first, int variant, uint8_t* dst, size_t dstep) {
    const int ry = variant ? 4899 : 9798, gy = variant ? 9617 : 19235, by = variant ? 1868 : 3735, shift = variant ? 14 : 15;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* p = src + (size_t)y * sstep + (size_t)x * cn;
            const int r = red_first ? p[0] : p[2], g = p[1], b = red_first ? p[2] : p[0];
            dst[(size_t)y * dstep + x] = (uint8_t)((r * ry + g * gy + b * by + (1 << (shift - 1))) >> shift);
        }
}

// ---- cv::undistortPoints(src, dst, K, distCoeffs, R = empty, P = K) for CV_32FC2 points (Frame::UndistortKeyPoints, src/Frame.cc:1003-1034) ----
// OpenCV calib3d/undistort (cvUndistortPointsInternal), the (k1, k2, p1, p2[, k3]) model the reference configures: everything in double,
// x = (u - cx) / fx with ifx = 1. / fx, five fixed-point iterations (the public overload's TermCriteria(MAX_ITER, 5, 0.01): count only)
//   r2 = x x + y y; icdist = (1 + ((k[7] r2 + k[6]) r2 + k[5]) r2) / (1 + ((k[4] r2 + k[1]) r2 + k[0]) r2)      (k[5..7] = 0 here: numerator 1)
//   dX = 2 k[2] x y + k[3] (r2 + 2 x x); dY = k[2] (r2 + 2 y y) + 2 k[3] x y; x = (x0 - dX) icdist; y = (y0 - dY) icdist
// (OpenCV >= 3.4.2 leaves the loop with the undistorted start value when icdist < 0: variant 0; 3.2 does not: variant 1), then the new
// projection xx = P00 x + P01 y + P02, yy = ..., ww = 1 / (P20 x + P21 y + P22) with P = K, and the result is stored as float.
static inline void undistort_points_f32(const float* src, int n, const float K[4], const float* dist, int ndist, int variant, float* dst) {
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], ifx = 1. / fx, ify = 1. / fy;
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ndist && i < 5; i++) k[i] = dist[i];
    for (int i = 0; i < n; i++) {
        const double u = src[2 * i], v = src[2 * i + 1];
        double x = (u - cx) * ifx, y = (v - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (variant == 0 && icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist; y = (y0 - deltaY) * icdist;
        }
        const double xx = fx * x + 0.0 * y + cx, yy = 0.0 * x + fy * y + cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
        dst[2 * i] = (float)(xx * ww); dst[2 * i + 1] = (float)(yy * ww);
    }
}

// ---- cv::copyMakeBorder(BORDER_REFLECT_101) ---------------------------------------------------
// dst is (w+left+right) x (h+top+bottom); src may alias the interior of dst.
static inline void make_border_reflect101(const uint8_t* src, int w, int h, size_t sstep,
                                          uint8_t* dst, size_t dstep, int top, int bottom, int left, int right) {
    const int dw = w + left + right;
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstep;
        uint8_t* D = dst + (size_t)(y + top) * dstep;
        if (D + left != S) memmove(D + left, S, w);
        for (int x = 0; x < left; x++) D[x] = D[left + reflect101(x - left, w)];
        for (int x = 0; x < right; x++) D[left + w + x] = D[left + reflect101(w + x, w)];
    }
    for (int y = 0; y < top; y++)
        memcpy(dst + (size_t)y * dstep, dst + (size_t)(top + reflect101(y - top, h)) * dstep, dw);
    for (int y = 0; y < bottom; y++)
        memcpy(dst + (size_t)(top + h + y) * dstep, dst + (size_t)(top + reflect101(h + y, h)) * dstep, dw);
}

// ---- cv::FAST, TYPE_9_16, with the OpenCV corner score and 3x3 strict non-max suppression -----
static const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// Is the pixel a FAST-9 corner at `threshold`?  (>= 9 contiguous ring pixels all brighter than
// v+threshold or all darker than v-threshold.)
static inline bool fast_is_corner(const uint8_t* p, size_t step, int threshold) {
    const int v = p[0];
    // Exact quick rejection (same idea as OpenCV's table test): a 9-arc contains at least one pixel
    // of every opposite pair (k, k+8), so the AND over pairs of the OR of the two classes must be !=0.
    int d = 3;
    for (int k = 0; k < 8 && d; k++) {
        int a = p[(ptrdiff_t)kRingDy[k] * (ptrdiff_t)step + kRingDx[k]];
        int b = p[(ptrdiff_t)kRingDy[k + 8] * (ptrdiff_t)step + kRingDx[k + 8]];
        int ca = (a < v - threshold ? 1 : 0) | (a > v + threshold ? 2 : 0);
        int cb = (b < v - threshold ? 1 : 0) | (b > v + threshold ? 2 : 0);
        d &= (ca | cb);
    }
    if (!d) return false;
    int run_d = 0, run_b = 0;
    for (int k = 0; k < 16 + 8; k++) {
        int x = p[(ptrdiff_t)kRingDy[k & 15] * (ptrdiff_t)step + kRingDx[k & 15]];
        if (x < v - threshold) { if (++run_d > 8) return true; } else run_d = 0;
        if (x > v + threshold) { if (++run_b > 8) return true; } else run_b = 0;
    }
    return false;
}
// OpenCV cornerScore<16>: largest threshold for which the pixel is still a corner
// (= max(threshold, max over the sixteen 9-arcs of min |v - ring|) - 1).
static inline int fast_corner_score(const uint8_t* p, size_t step, int threshold) {
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[(ptrdiff_t)kRingDy[k & 15] * (ptrdiff_t)step + kRingDx[k & 15]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]); a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]); a = std::min(a, d[k + 5]); a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]); a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]); b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]); b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]); b = std::max(b, d[k + 7]); b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}
struct FastPoint { int x, y, score; };
static inline void fast9_16(const uint8_t* img, int w, int h, size_t step, int threshold, bool nonmax,
                            std::vector<FastPoint>& out) {
    out.clear();
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    if (w < 7 || h < 7) return;
    std::vector<uint8_t> sc((size_t)w * h, 0);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * step + x;
            if (fast_is_corner(p, step, threshold)) {
                if (nonmax) sc[(size_t)y * w + x] = (uint8_t)fast_corner_score(p, step, threshold);
                else out.push_back({x, y, 0});
            }
        }
    if (!nonmax) return;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const int s = sc[(size_t)y * w + x];
            if (!s) continue;   // a corner's score is >= threshold; threshold 0 corners of score 0 are
                                // never "> neighbours" anyway
            const uint8_t* q = &sc[(size_t)y * w + x];
            if (s > q[-1] && s > q[1] && s > q[-w - 1] && s > q[-w] && s > q[-w + 1] &&
                s > q[w - 1] && s > q[w] && s > q[w + 1])
                out.push_back({x, y, s});
        }
}

// ---- cv::GaussianBlur(Size(7,7), 2, 2, BORDER_REFLECT_101) on 8UC1 ------------------------------
// taps_variant 0: OpenCV >= 3.4.x/4.x bit-exact fixed-point path  [18,34,48,56,48,34,18]/256
// taps_variant 1: OpenCV 3.2-era integer separable filter          [18,34,49,55,49,34,18]/256
static inline const int* gauss7_taps(int variant) {
    static const int A[7] = {18, 34, 48, 56, 48, 34, 18};
    static const int B[7] = {18, 34, 49, 55, 49, 34, 18};
    return variant == 1 ? B : A;
}
static inline void gaussian_blur7_u8(const uint8_t* src, int w, int h, size_t sstep,
                                     uint8_t* dst, size_t dstep, int taps_variant) {
    const int* k = gauss7_taps(taps_variant);
    std::vector<uint32_t> H((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int i = 0; i < 7; i++) s += k[i] * src[(size_t)y * sstep + reflect101(x + i - 3, w)];
            H[(size_t)y * w + x] = s;
        }
    std::vector<uint8_t> out((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int j = 0; j < 7; j++) s += k[j] * H[(size_t)reflect101(y + j - 3, h) * w + x];
            uint32_t v = (s + 32768u) >> 16;
            out[(size_t)y * w + x] = (uint8_t)(v > 255 ? 255 : v);
        }
    for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * dstep, &out[(size_t)y * w], w);
}

// ---- cv::fastAtan2(y, x): degrees in [0,360], fp32 polynomial, no FMA contraction ---------------
// (pinned even inside a translation unit built with the reference's own -O3 -march=native, oracle/Makefile _ref/libref_orb_native.so: OpenCV is a
// separately built library, the flags of the reference's CMakeLists.txt do not reach it)
#if defined(__GNUC__) && !defined(__clang__)
__attribute__((optimize("fp-contract=off")))
#endif
static inline float fast_atan2_deg(float y, float x) {
    const float s = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float eps = (float)DBL_EPSILON;
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps); c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps); c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---- cv::norm(A, B, NORM_L1) on two 8U patches ---------------------------------------------------
static inline double norm_l1_u8(const uint8_t* a, size_t astep, const uint8_t* b, size_t bstep, int w, int h) {
    long s = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) s += std::abs((int)a[y * astep + x] - (int)b[y * bstep + x]);
    return (double)s;
}

}  // namespace orbp
