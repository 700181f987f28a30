// What a caller of the reference sees per frame: ORBextractor::operator() on one 8-bit image, called again and again on one instance
// (src/Frame.cc:1038 ExtractORB).  Built like facade_driver.cpp: against the reference's own header + source (CPU), or against the drop-in
// facade include/orb_slam3_amd/ORBextractor.h + liborbx_hip.so.  Prints the mean wall time per call.
//   facade_latency in.raw w h nfeatures calls [export_pyramid = 1]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ORBextractor.h"

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: facade_latency in.raw w h nfeatures calls [export_pyramid]\n"); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]), nf = atoi(argv[4]), calls = atoi(argv[5]);
    const int exportp = argc > 6 ? atoi(argv[6]) : 1;
    std::vector<unsigned char> buf((size_t)w * h);
    FILE* f = fopen(argv[1], "rb"); if (!f || fread(buf.data(), 1, buf.size(), f) != buf.size()) { fprintf(stderr, "read failed\n"); return 3; } fclose(f);
    cv::Mat im(h, w, CV_8UC1, buf.data());
    ORB_SLAM3::ORBextractor ex(nf, 1.2f, 8, 20, 7);
#ifdef ORBX_FACADE
    ex.SetExportPyramid(exportp != 0);
#else
    (void)exportp;
#endif
    std::vector<cv::KeyPoint> kps; cv::Mat desc;
    std::vector<int> lap = {0, 0};
    for (int i = 0; i < 3; i++) ex(im, cv::Mat(), kps, desc, lap);          // first use allocates
    const auto t0 = std::chrono::steady_clock::now();
    size_t total = 0;
    for (int i = 0; i < calls; i++) { ex(im, cv::Mat(), kps, desc, lap); total += kps.size(); }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / calls;
    printf("%dx%d nfeatures %d: %.3f ms per call, %.0f keypoints per call%s\n", w, h, nf, ms, (double)total / calls,
#ifdef ORBX_FACADE
           exportp ? " (mvImagePyramid exported)" : " (mvImagePyramid not exported)"
#else
           " (reference source on this host, one thread)"
#endif
    );
    return 0;
}
