// oracle/opencv_shim/opencv2/opencv.hpp — TEST INFRASTRUCTURE ONLY.
//
// A minimal stand-in for the slice of OpenCV that the reference's src/ORBextractor.cc and
// include/ORBextractor.h use (includes at ORBextractor.cc:54-57, ORBextractor.h:24).  OpenCV is not
// installed in this image and cannot be (no network), so to run the reference's *own* extractor
// source as the oracle we compile it, unmodified and read in place from /root/reference, against
// this header (oracle/Makefile -> oracle/_ref/).  The same header lets tests compile the product's
// C++ facade (include/orb_slam3_amd/ORBextractor.h) without OpenCV.
//
// The arithmetic of cv::resize / cv::FAST / cv::GaussianBlur / cv::fastAtan2 / cvRound lives in
// ../../orb_primitives.h and is a restatement of OpenCV's scalar algorithms — PARITY UNPINNED
// against a real OpenCV (see that header).
#pragma once
#include <algorithm>
#include <cassert>
#include <climits>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <sstream>
#include <iostream>
#include <fstream>
#include <vector>
#include "../../orb_primitives.h"

#define CV_PI 3.1415926535897932384626433832795
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5

typedef unsigned char uchar;

static inline int cvRound(double v) { return orbp::round_half_even(v); }
static inline int cvFloor(double v) { return orbp::floor_i(v); }
static inline int cvCeil(double v) { return orbp::ceil_i(v); }

namespace cv {

// Gaussian taps variant used by GaussianBlur below (orb_primitives.h gauss7_taps); settable by
// the test drivers so that both OpenCV generations can be exercised.
inline int& shim_gauss_variant() { static int v = 0; return v; }

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T _x, T _y) : x(_x), y(_y) {}
};
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;
struct Point3f { float x, y, z; Point3f() : x(0), y(0), z(0) {} Point3f(float a, float b, float c) : x(a), y(b), z(c) {} };
static inline Point2f& operator*=(Point2f& p, float s) { p.x = p.x * s; p.y = p.y * s; return p; }

struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x, y, width, height; Rect() : x(0), y(0), width(0), height(0) {} Rect(int _x, int _y, int w, int h) : x(_x), y(_y), width(w), height(h) {} };

struct KeyPoint {
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1)
        : pt(x, y), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id) {}
};

enum { INTER_LINEAR = 1 };
enum { BORDER_REFLECT_101 = 4, BORDER_DEFAULT = 4, BORDER_ISOLATED = 16 };
enum { NORM_L1 = 2, NORM_HAMMING = 6 };

struct MatStep {
    size_t v; MatStep() : v(0) {} MatStep(size_t s) : v(s) {}
    operator size_t() const { return v; }
};

class Mat {
public:
    int rows, cols; uchar* data; MatStep step;
    Mat() : rows(0), cols(0), data(nullptr), step(0), type_(CV_8UC1) {}
    Mat(int r, int c, int type) : Mat() { create(r, c, type); }
    Mat(Size s, int type) : Mat() { create(s.height, s.width, type); }
    // user-data constructor (no ownership)
    Mat(int r, int c, int type, void* d, size_t st = 0) : rows(r), cols(c), data((uchar*)d), step(st ? st : (size_t)c * esz(type)), type_(type) {}
    static Mat eye(int r, int c, int type) { Mat m = zeros(r, c, type); for (int i = 0; i < r && i < c; i++) m.at<float>(i, i) = 1.0f; return m; }
    static Mat zeros(int r, int c, int type) { Mat m(r, c, type); if (m.data) memset(m.data, 0, (size_t)r * m.step); return m; }
    void create(int r, int c, int type) {
        if (data && r == rows && c == cols && type == type_) return;
        type_ = type; rows = r; cols = c; step = (size_t)c * esz(type);
        buf_ = std::shared_ptr<std::vector<uchar>>(new std::vector<uchar>((size_t)r * step + 64));
        data = buf_->data();
    }
    void create(Size s, int type) { create(s.height, s.width, type); }
    void release() { buf_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    Size size() const { return Size(cols, rows); }
    size_t step1() const { return step / esz1(type_); }
    size_t elemSize() const { return esz(type_); }
    Mat operator()(const Rect& r) const { Mat m(*this); m.data = data + (size_t)r.y * step + (size_t)r.x * esz(type_); m.rows = r.height; m.cols = r.width; return m; }
    Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
    Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
    Mat row(int y) const { return rowRange(y, y + 1); }
    Mat clone() const { Mat m(rows, cols, type_); for (int y = 0; y < rows; y++) memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * esz(type_)); return m; }
    void copyTo(Mat& dst) const { dst.create(rows, cols, type_); for (int y = 0; y < rows; y++) memcpy(dst.data + (size_t)y * dst.step, data + (size_t)y * step, (size_t)cols * esz(type_)); }
    void copyTo(Mat&& dst) const { copyTo(dst); }   // e.g. desc.row(i).copyTo(descriptors.row(j))
    template <typename T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step + (size_t)x * sizeof(T)); }
    template <typename T> const T& at(int y, int x) const { return *(const T*)(data + (size_t)y * step + (size_t)x * sizeof(T)); }
    template <typename T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <typename T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    size_t total() const { return (size_t)rows * cols; }
#ifdef ORBX_TRACKING_WORLD      // (oracle/slam_shim/tracking_world.h: compile check of Tracking.cc; declared only, nothing of it is linked)
    int channels() const; void resize(size_t rows_); void convertTo(Mat& m, int rtype, double alpha = 1, double beta = 0) const;
#endif
    Mat reshape(int /*cn*/) const { return *this; }             // only reached with lens distortion, which the oracle never configures
    uchar* ptr(int y = 0) { return data + (size_t)y * step; }
    const uchar* ptr(int y = 0) const { return data + (size_t)y * step; }
    template <typename T> T* ptr(int y = 0) { return (T*)(data + (size_t)y * step); }
    template <typename T> const T* ptr(int y = 0) const { return (const T*)(data + (size_t)y * step); }
private:
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    static size_t esz1(int type) { return esz(type); }
    int type_;
    std::shared_ptr<std::vector<uchar>> buf_;
};

class _InputArray {
public:
    _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
    bool empty() const { return m_->empty(); }
    Mat getMat() const { return *m_; }
protected:
    Mat* m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray(Mat& m) : _InputArray(m) {}
    void create(int r, int c, int type) const { m_->create(r, c, type); }
    void release() const { m_->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

// What src/CameraModels/{Pinhole,KannalaBrandt8}.cpp name to compile: (cv::Mat_<float>(r, c) << a, b, ...) for toK() (:255-258 / :296-302),
// Mat::eye is below, cv::fisheye::undistortPoints only inside ReconstructWithTwoViews (monocular initialisation: out of scope, never called).
template <typename T> class Mat_ : public Mat {
public:
    Mat_(int r, int c) : Mat(r, c, CV_32F) { static_assert(sizeof(T) == 4, "float matrices only"); }
};
template <typename T> struct MatCommaInitializer_ {
    Mat m; int k;
    MatCommaInitializer_& operator,(T v) { m.at<T>(k / m.cols, k % m.cols) = v; k++; return *this; }
    operator Mat() const { return m; }
};
template <typename T> MatCommaInitializer_<T> operator<<(const Mat_<T>& m, T v) { MatCommaInitializer_<T> ci{m, 0}; ci, v; return ci; }
namespace fisheye {
static inline void undistortPoints(const std::vector<Point2f>&, std::vector<Point2f>&, const Mat&, const Mat&, const Mat&, const Mat&) { abort(); }
}

static inline float fastAtan2(float y, float x) { return orbp::fast_atan2_deg(y, x); }

static inline void FAST(const Mat& img, std::vector<KeyPoint>& kps, int threshold, bool nonmax = true) {
    std::vector<orbp::FastPoint> pts;
    orbp::fast9_16(img.data, img.cols, img.rows, img.step, threshold, nonmax, pts);
    kps.clear();
    for (const auto& p : pts) kps.push_back(KeyPoint((float)p.x, (float)p.y, 7.f, -1, (float)p.score));
}

static inline void resize(const Mat& src, Mat& dst, Size dsize, double = 0, double = 0, int = INTER_LINEAR) {
    dst.create(dsize, src.type());
    orbp::resize_linear_u8(src.data, src.cols, src.rows, src.step, dst.data, dst.cols, dst.rows, dst.step);
}

static inline void copyMakeBorder(const Mat& src, Mat& dst, int top, int bottom, int left, int right, int /*borderType*/) {
    dst.create(src.rows + top + bottom, src.cols + left + right, src.type());
    orbp::make_border_reflect101(src.data, src.cols, src.rows, src.step, dst.data, dst.step, top, bottom, left, right);
}

static inline void GaussianBlur(const Mat& src, Mat& dst, Size ksize, double sx, double sy = 0, int = BORDER_DEFAULT) {
    assert(ksize.width == 7 && ksize.height == 7 && sx == 2.0 && (sy == 2.0 || sy == 0.0));
    (void)ksize; (void)sx; (void)sy;
    dst.create(src.rows, src.cols, src.type());
    orbp::gaussian_blur7_u8(src.data, src.cols, src.rows, src.step, dst.data, dst.step, shim_gauss_variant());
}

static inline double norm(const Mat& a, const Mat& b, int /*NORM_L1*/) {
    return orbp::norm_l1_u8(a.data, a.step, b.data, b.step, a.cols, a.rows);
}

// Names Frame.cc needs to compile.  undistortPoints (src/Frame.cc:1019, :1061; CV_32FC2 points in place, R empty, P = K) is the restated
// primitive of orb_primitives.h; vconcat / BFMatcher serve the fisheye-rig constructor (:1514, :1553): brute-force Hamming 2-NN with strict '<'
// insertion, equal distances keep the lower train index first (OpenCV's BFMatcher order; restated, OpenCV is not installed).
inline int& shim_undistort_variant() { static int v = 0; return v; }
static inline void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& dist, const Mat&, const Mat&) {
    const float k4[4] = {K.at<float>(0, 0), K.at<float>(1, 1), K.at<float>(0, 2), K.at<float>(1, 2)};
    float d[5] = {0, 0, 0, 0, 0};
    const int nd = dist.rows * dist.cols;
    for (int i = 0; i < nd && i < 5; i++) d[i] = ((const float*)dist.data)[i];
    std::vector<float> out((size_t)2 * src.rows + 2);
    orbp::undistort_points_f32((const float*)src.data, src.rows, k4, d, nd < 5 ? nd : 5, shim_undistort_variant(), out.data());
    if (dst.data != src.data) dst.create(src.rows, src.cols, src.type());
    memcpy(dst.data, out.data(), sizeof(float) * 2 * (size_t)src.rows);
}
static inline void vconcat(const Mat& a, const Mat& b, Mat& dst) {
    Mat r(a.rows + b.rows, a.cols, CV_8UC1);
    for (int y = 0; y < a.rows; y++) memcpy(r.ptr(y), a.ptr(y), (size_t)a.cols);
    for (int y = 0; y < b.rows; y++) memcpy(r.ptr(a.rows + y), b.ptr(y), (size_t)b.cols);
    dst = r;
}
struct DMatch {
    int queryIdx, trainIdx, imgIdx; float distance;
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.4e38f) {}
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(0), distance(d) {}
};
class BFMatcher {
public:
    BFMatcher(int /*normType*/ = NORM_HAMMING, bool /*crossCheck*/ = false) {}
    void knnMatch(const Mat& query, const Mat& train, std::vector<std::vector<DMatch>>& matches, int k) const {
        matches.assign(query.rows, std::vector<DMatch>());
        for (int i = 0; i < query.rows; i++) {
            std::vector<std::pair<int, int>> best;                  // (distance, train index), ascending, at most k
            for (int j = 0; j < train.rows; j++) {
                int d = 0;
                for (int b = 0; b < query.cols; b++) d += __builtin_popcount((unsigned)(query.ptr(i)[b] ^ train.ptr(j)[b]));
                size_t pos = best.size();
                while (pos > 0 && d < best[pos - 1].first) pos--;
                if ((int)pos < k) { best.insert(best.begin() + pos, std::make_pair(d, j)); if ((int)best.size() > k) best.pop_back(); }
            }
            for (size_t n = 0; n < best.size(); n++) matches[i].push_back(DMatch(i, best[n].second, (float)best[n].first));
        }
    }
};

struct KeyPointsFilter {   // referenced only by the reference's dead ComputeKeyPointsOld
    static void retainBest(std::vector<KeyPoint>& kps, int n) {
        if (n >= 0 && (int)kps.size() > n) {
            std::stable_sort(kps.begin(), kps.end(), [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
            kps.resize(n);
        }
    }
};

// cv::FileStorage / cv::FileNode: named by DBoW2's TemplatedVocabulary.h (YAML save/load, virtual members that must compile).
// The oracle loads vocabularies with the reference's own loadFromTextFile, so these are never-opened stubs.
class FileNode {
public:
    FileNode operator[](const char*) const { return FileNode(); }
    FileNode operator[](const std::string&) const { return FileNode(); }
    FileNode operator[](int) const { return FileNode(); }
    size_t size() const { return 0; }
    operator int() const { return 0; }
    operator double() const { return 0.0; }
    operator std::string() const { return std::string(); }
#ifdef ORBX_TRACKING_WORLD
    bool empty() const; bool isReal() const; bool isInt() const; bool isString() const; double real() const; Mat mat() const; operator float() const;
    template <typename T> T operator>>(T&) const;
#endif
};
class FileStorage {
public:
    enum { READ = 0, WRITE = 1 };
    FileStorage(const char*, int) {}
    FileStorage(const std::string&, int) {}
    bool isOpened() const { return false; }
    void release() {}
    FileNode operator[](const char*) const { return FileNode(); }
    FileNode operator[](const std::string&) const { return FileNode(); }
};
template <typename T> static inline FileStorage& operator<<(FileStorage& fs, const T&) { return fs; }
#ifdef ORBX_TRACKING_WORLD
enum { COLOR_BGR2GRAY = 6, COLOR_RGB2GRAY = 7, COLOR_BGRA2GRAY = 10, COLOR_RGBA2GRAY = 11 };
void cvtColor(InputArray src, OutputArray dst, int code, int dstCn = 0);
std::ostream& operator<<(std::ostream& os, const Mat& m);
#endif

}  // namespace cv
