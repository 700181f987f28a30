// oracle/ref_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// C-ABI wrapper around the REFERENCE's own ORB_SLAM3::ORBextractor, compiled from
// /root/reference/src/ORBextractor.cc (unmodified, read in place; never copied into this repo)
// against oracle/opencv_shim.  Built by oracle/Makefile into oracle/_ref/libref_orb.so, which is
// git-ignored but travels to the GPU box.  Used by tests/ and bench.py's cpu_baseline leg only.
#include <cstdint>
#include <cstring>
#include <vector>
#include "ORBextractor.h"   // the reference header, via -I/root/reference/include

namespace {
// Derived class only to reach the reference's protected stage functions (include/ORBextractor.h:86-92).
struct Probe : public ORB_SLAM3::ORBextractor {
    using ORB_SLAM3::ORBextractor::ORBextractor;
    void pyramid(const cv::Mat& im) { ComputePyramid(im); }
    void keypoints(std::vector<std::vector<cv::KeyPoint>>& all) { ComputeKeyPointsOctTree(all); }
    const std::vector<int>& quotas() const { return mnFeaturesPerLevel; }
    const std::vector<int>& umax_tab() const { return umax; }
    const std::vector<cv::Point>& pattern_tab() const { return pattern; }
};
struct RefKp { float x, y, size, angle, response; int octave, class_id; };
}  // namespace

extern "C" {

void* ref_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    return new Probe(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void ref_orb_destroy(void* h) { delete (Probe*)h; }

// Full operator() (src/ORBextractor.cc:1557).  kps: cap x 28 B, desc: cap x 32 B.
// Returns monoIndex (or -1), *n_out = number of keypoints (may exceed cap: then nothing is copied).
int ref_orb_extract(void* h, const uint8_t* img, int w, int hgt, int stride, int lap0, int lap1,
                    int gauss_variant, void* kps_out, uint8_t* desc_out, int cap, int* n_out) {
    Probe* p = (Probe*)h;
    cv::shim_gauss_variant() = gauss_variant;
    cv::Mat im(hgt, w, CV_8UC1, (void*)img, (size_t)stride);
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    std::vector<int> lap = {lap0, lap1};
    int mono = (*p)(im, cv::Mat(), kps, desc, lap);
    *n_out = (int)kps.size();
    if ((int)kps.size() <= cap) {
        RefKp* o = (RefKp*)kps_out;
        for (size_t i = 0; i < kps.size(); i++) {
            o[i] = {kps[i].pt.x, kps[i].pt.y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave, kps[i].class_id};
            memcpy(desc_out + 32 * i, desc.ptr((int)i), 32);
        }
    }
    return mono;
}

// Un-bordered level image of the last call (mvImagePyramid[level], include/ORBextractor.h:83).
int ref_orb_pyramid_level(void* h, int level, uint8_t* dst, int cap, int* w, int* hgt) {
    Probe* p = (Probe*)h;
    const cv::Mat& m = p->mvImagePyramid[level];
    *w = m.cols; *hgt = m.rows;
    if ((long)m.cols * m.rows > cap) return -1;
    for (int y = 0; y < m.rows; y++) memcpy(dst + (size_t)y * m.cols, m.ptr(y), m.cols);
    return 0;
}

// Stage probe: ComputePyramid + ComputeKeyPointsOctTree only (quadtree + orientation, level coords).
// counts[nlevels]; kps_out holds levels back to back.
int ref_orb_keypoints_per_level(void* h, const uint8_t* img, int w, int hgt, int stride,
                                void* kps_out, int cap, int* counts) {
    Probe* p = (Probe*)h;
    cv::Mat im(hgt, w, CV_8UC1, (void*)img, (size_t)stride);
    p->pyramid(im);
    std::vector<std::vector<cv::KeyPoint>> all;
    p->keypoints(all);
    RefKp* o = (RefKp*)kps_out;
    int n = 0;
    for (size_t l = 0; l < all.size(); l++) {
        counts[l] = (int)all[l].size();
        for (const auto& k : all[l]) {
            if (n < cap) o[n] = {k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave, k.class_id};
            n++;
        }
    }
    return n;
}

// Constant tables of the reference constructor (src/ORBextractor.cc:468-571) for known-answer tests.
void ref_orb_tables(void* h, int* quotas, int* umax16, int* pattern1024, float* scale, float* inv_scale,
                    float* sigma2, float* inv_sigma2) {
    Probe* p = (Probe*)h;
    int nl = p->GetLevels();
    for (int i = 0; i < nl; i++) quotas[i] = p->quotas()[i];
    for (int i = 0; i < 16; i++) umax16[i] = p->umax_tab()[i];
    for (int i = 0; i < 512; i++) { pattern1024[2 * i] = p->pattern_tab()[i].x; pattern1024[2 * i + 1] = p->pattern_tab()[i].y; }
    std::vector<float> a = p->GetScaleFactors(), b = p->GetInverseScaleFactors(), c = p->GetScaleSigmaSquares(), d = p->GetInverseScaleSigmaSquares();
    for (int i = 0; i < nl; i++) { scale[i] = a[i]; inv_scale[i] = b[i]; sigma2[i] = c[i]; inv_sigma2[i] = d[i]; }
}

}  // extern "C"
