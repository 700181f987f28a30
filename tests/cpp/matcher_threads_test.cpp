// Three threads use the ORBmatcher facade at the same time, as Tracking, LocalMapping and LoopClosing do in the reference:
// every thread owns a library handle (thread_local in the facade), so the calls run concurrently; each thread's results must equal the
// results of the same calls made one after the other.  Input: one raw image.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>
#include "ORBextractor.h"
#include "ORBmatcher.h"

struct MockMapPoint {
    bool mbTrackInView = true, mbTrackInViewR = false; float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackProjYR = 0, mTrackViewCos = 1, mTrackViewCosR = 1, mTrackDepth = 1;
    int mnTrackScaleLevel = 0, mnTrackScaleLevelR = -1; bool bad = false; int nobs = 1; cv::Mat desc;
    bool isBad() { return bad; } int Observations() { return nobs; } cv::Mat GetDescriptor() { return desc; }
};
struct MockFrame {
    int N = 0, Nleft = -1; std::vector<cv::KeyPoint> mvKeysUn, mvKeys, mvKeysRight; std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch; cv::Mat mDescriptors; std::vector<float> mvuRight, mvDepth;
    std::vector<MockMapPoint*> mvpMapPoints; float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0, mbf = 0, mb = 0;
    std::vector<float> mvScaleFactors;
};

static std::vector<unsigned char> read_raw(const char* p, size_t n) { std::vector<unsigned char> b(n); FILE* f = fopen(p, "rb"); if (!f || fread(b.data(), 1, n, f) != n) { fprintf(stderr, "read %s failed\n", p); exit(3); } fclose(f); return b; }

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    std::vector<unsigned char> bl = read_raw(argv[1], (size_t)w * h);
    cv::Mat im(h, w, CV_8UC1, bl.data());
    ORB_SLAM3::ORBextractor ex(500, 1.2f, 8, 20, 7);
    MockFrame base;
    std::vector<int> lap = {0, 0};
    ex(im, cv::Mat(), base.mvKeysUn, base.mDescriptors, lap);
    base.N = (int)base.mvKeysUn.size(); base.mnMaxX = (float)w; base.mnMaxY = (float)h;
    base.mfGridElementWidthInv = 64.f / w; base.mfGridElementHeightInv = 48.f / h; base.mvScaleFactors = ex.GetScaleFactors();
    base.mvuRight.assign(base.N, -1.0f);
    const int T = 3, REPS = 4;
    // per thread: its own frame copy, its own local map, the expected assignment (computed sequentially first)
    std::vector<MockFrame> frames(T, base);
    std::vector<std::vector<MockMapPoint>> maps(T);
    std::vector<std::vector<MockMapPoint*>> vps(T);
    std::vector<std::vector<int>> expect(T);
    std::vector<std::vector<int>> dist_expect(T);
    std::vector<cv::Mat> dA(T), dB(T);
    for (int t = 0; t < T; t++) {
        std::mt19937 rng(100 + t);
        maps[t].resize(800 + 300 * t);
        for (auto& p : maps[t]) {
            const int s = rng() % base.N;
            p.mTrackProjX = base.mvKeysUn[s].pt.x + (int)(rng() % 5) - 2; p.mTrackProjY = base.mvKeysUn[s].pt.y + (int)(rng() % 5) - 2;
            p.mnTrackScaleLevel = base.mvKeysUn[s].octave; p.mTrackViewCos = (rng() & 1) ? 0.9985f : 0.99f; p.nobs = (rng() % 10) ? 2 : 0;
            p.desc = base.mDescriptors.row(s).clone();
            for (int b = 0; b < (int)(rng() % 30); b++) p.desc.ptr(0)[rng() % 32] ^= (unsigned char)(1 << (rng() % 8));
            vps[t].push_back(&p);
        }
        dA[t].create(40 + t, 32, CV_8U); dB[t].create(50, 32, CV_8U);
        for (int i = 0; i < dA[t].rows * 32; i++) dA[t].ptr(0)[i] = (unsigned char)rng();
        for (int i = 0; i < dB[t].rows * 32; i++) dB[t].ptr(0)[i] = (unsigned char)rng();
    }
    // optional 4th argument: number of GPUs D.  Thread t then works on GPU t % D - its matcher handle through ORBmatcher::SetThreadDevice, and
    // (second part) an extractor of its own with deviceId = t % D plus the all-gather of the descriptor blocks over the C ABI, one rank per GPU.
    const int D = argc > 4 ? atoi(argv[4]) : 1;
    auto work = [&](int t, std::vector<int>& assign, std::vector<int>& dist) {
        ORB_SLAM3::ORBmatcher::SetThreadDevice(t % D);
        ORB_SLAM3::ORBmatcher m(0.8f);
        MockFrame& F = frames[t];
        F.mvpMapPoints.assign(F.N, nullptr);
        m.SearchByProjection(F, vps[t], 3.0f);
        assign.resize(F.N);
        for (int i = 0; i < F.N; i++) assign[i] = F.mvpMapPoints[i] ? (int)(F.mvpMapPoints[i] - maps[t].data()) : -1;
        ORB_SLAM3::ORBmatcher::DescriptorDistances(dA[t], dB[t], dist);
        for (int i = 0; i < dA[t].rows; i++)         // the single-pair form against the batched one
            if (ORB_SLAM3::ORBmatcher::DescriptorDistance(dA[t].row(i), dB[t].row(i % dB[t].rows)) != dist[(size_t)i * dB[t].rows + i % dB[t].rows]) dist[0] = -1;
    };
    for (int t = 0; t < T; t++) work(t, expect[t], dist_expect[t]);
    std::atomic<int> bad(0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            for (int r = 0; r < REPS; r++) {
                std::vector<int> a, d;
                work(t, a, d);
                if (a != expect[t] || d != dist_expect[t] || d[0] < 0) bad++;
            }
        });
    for (auto& x : th) x.join();
    int matched = 0; for (int v : expect[0]) matched += v >= 0;
    printf("threads=%d reps=%d devices=%d mismatches=%d matches(thread 0)=%d\n", T, REPS, D, bad.load(), matched);
    // ---- one rank per GPU: extractor + matcher on device r, the descriptor exchange of BASELINE.json configs[4] from C++ (no Python, no torch) ----
    std::atomic<int> bad2(0);
    if (D > 1) {
        uint8_t id[ORBX_COMM_ID_BYTES];
        if (orbx_comm_unique_id(id) != ORBX_OK) { fprintf(stderr, "unique id: %s\n", orbx_last_error()); return 1; }
        std::vector<std::thread> ranks;
        for (int r = 0; r < D; r++)
            ranks.emplace_back([&, r]() {
                ORB_SLAM3::ORBmatcher::SetThreadDevice(r);
                ORB_SLAM3::ORBextractor exr(500, 1.2f, 8, 20, 7, /*deviceId*/ r);
                if (orbx_device_id(exr.Handle()) != r && orbx_device_id(exr.Handle()) != ORBX_DEVICE_HOST) bad2++;
                orbx_comm* c = nullptr;
                if (orbx_comm_create(&c, D, r, id, r) != ORBX_OK) { fprintf(stderr, "rank %d: %s\n", r, orbx_last_error()); bad2++; return; }
                // every rank extracts its own image: the base image, mirrored top to bottom on odd ranks
                std::vector<unsigned char> br(bl);
                if (r & 1) for (int y = 0; y < h; y++) memcpy(br.data() + (size_t)y * w, bl.data() + (size_t)(h - 1 - y) * w, w);
                cv::Mat imr(h, w, CV_8UC1, br.data());
                std::vector<cv::KeyPoint> kr; cv::Mat dr; std::vector<int> lp = {0, 0};
                exr(imr, cv::Mat(), kr, dr, lp);
                if (r == 0 && (kr.size() != base.mvKeysUn.size() || memcmp(dr.data, base.mDescriptors.data, (size_t)dr.rows * 32))) bad2++;   // GPU r == GPU 0's answer
                int B = 0, cap = 0;
                if (orbx_allgather_descriptors(exr.Handle(), c, nullptr, nullptr, &B, &cap) != ORBX_OK) { fprintf(stderr, "rank %d: %s\n", r, orbx_last_error()); bad2++; orbx_comm_destroy(c); return; }
                std::vector<uint8_t> all((size_t)D * B * cap * 32); std::vector<int> nall((size_t)D * B);
                if (orbx_comm_fetch(c, all.data(), nall.data()) != ORBX_OK || B != 1) { bad2++; orbx_comm_destroy(c); return; }
                if (nall[r] != dr.rows || memcmp(all.data() + (size_t)r * cap * 32, dr.data, (size_t)dr.rows * 32)) bad2++;                    // my block at my rank
                if (nall[0] != (int)base.mvKeysUn.size() || memcmp(all.data(), base.mDescriptors.data, (size_t)nall[0] * 32)) bad2++;           // rank 0's block everywhere
                // the matcher of this thread searches on this GPU
                std::vector<int> a, d2; work(r, a, d2);
                if (a != expect[r]) bad2++;
                orbx_comm_destroy(c);
            });
        for (auto& x : ranks) x.join();
        printf("ranks=%d exchange mismatches=%d\n", D, bad2.load());
    }
    return (bad.load() == 0 && bad2.load() == 0 && matched > 50) ? 0 : 1;
}
