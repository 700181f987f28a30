// Compiles the ORBmatcher facade against mock Frame / MapPoint structs that carry the reference's member names
// (include/Frame.h, include/MapPoint.h), runs SearchByProjection + ComputeStereoMatches + DescriptorDistance through it and
// checks them against the oracle restatement (liborb_oracle.so).  Input: two raw images (left, right).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
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
struct OFrame { int N; const void* keys; const unsigned char* desc; const float* u_right; const unsigned char* occupied; float min_x, min_y, max_x, max_y, gw_inv, gh_inv, mbf; int nlevels; const float* scale; };
struct OMapPoints { int M; const unsigned char* in_view; const float* proj_x; const float* proj_y; const float* proj_xr; const int* scale_level; const float* view_cos; const float* track_depth; const unsigned char* is_bad; const unsigned char* has_obs; const unsigned char* desc; };
extern "C" int orbo_search_by_projection_mappoints(const OFrame*, const OMapPoints*, float, int, float, float, int*);
extern "C" int orbo_descriptor_distance(const unsigned char*, const unsigned char*);

static std::vector<unsigned char> read_raw(const char* p, size_t n) { std::vector<unsigned char> b(n); FILE* f = fopen(p, "rb"); if (!f || fread(b.data(), 1, n, f) != n) { fprintf(stderr, "read %s failed\n", p); exit(3); } fclose(f); return b; }

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int w = atoi(argv[3]), h = atoi(argv[4]);
    std::vector<unsigned char> bl = read_raw(argv[1], (size_t)w * h), br = read_raw(argv[2], (size_t)w * h);
    cv::Mat imL(h, w, CV_8UC1, bl.data()), imR(h, w, CV_8UC1, br.data());
    ORB_SLAM3::ORBextractor exL(500, 1.2f, 8, 20, 7), exR(500, 1.2f, 8, 20, 7);
    MockFrame F;
    std::vector<cv::KeyPoint> kR; cv::Mat dR; std::vector<int> lap = {0, 0};
    exL(imL, cv::Mat(), F.mvKeysUn, F.mDescriptors, lap);
    exR(imR, cv::Mat(), kR, dR, lap);
    F.N = (int)F.mvKeysUn.size(); F.mnMaxX = (float)w; F.mnMaxY = (float)h;
    F.mfGridElementWidthInv = 64.f / (F.mnMaxX - F.mnMinX); F.mfGridElementHeightInv = 48.f / (F.mnMaxY - F.mnMinY);
    F.mbf = 458.654f * 0.110074f; F.mb = 0.110074f; F.mvScaleFactors = exL.GetScaleFactors();
    ORB_SLAM3::ComputeStereoMatches(F, &exL, &exR);
    int nstereo = 0; for (float u : F.mvuRight) nstereo += u >= 0;
    // synthetic local map: noisy copies of the frame's own keypoints
    std::mt19937 rng(7);
    std::vector<MockMapPoint> mps(1500); std::vector<MockMapPoint*> vp;
    for (auto& p : mps) {
        const int s = rng() % F.N;
        p.mTrackProjX = F.mvKeysUn[s].pt.x + (int)(rng() % 5) - 2; p.mTrackProjY = F.mvKeysUn[s].pt.y + (int)(rng() % 5) - 2;
        p.mTrackProjXR = F.mvuRight[s] > 0 ? F.mvuRight[s] + 0.5f : p.mTrackProjX - 4; p.mnTrackScaleLevel = F.mvKeysUn[s].octave;
        p.mTrackViewCos = (rng() & 1) ? 0.9985f : 0.99f; p.mTrackDepth = 1 + (rng() % 50); p.bad = (rng() % 40) == 0; p.nobs = (rng() % 10) ? 2 : 0;
        p.mbTrackInView = (rng() % 10) != 0;
        p.desc = F.mDescriptors.row(s).clone();
        for (int b = 0; b < (int)(rng() % 30); b++) p.desc.ptr(0)[rng() % 32] ^= (unsigned char)(1 << (rng() % 8));
        vp.push_back(&p);
    }
    F.mvpMapPoints.assign(F.N, nullptr);
    // oracle on the same inputs
    std::vector<unsigned char> occ(F.N, 0), inV, bad, obs, md; std::vector<float> px, py, pxr, vc, dep; std::vector<int> lvl;
    struct K { float x, y, size, angle, response; int octave, class_id; }; std::vector<K> keys(F.N);
    for (int i = 0; i < F.N; i++) { const cv::KeyPoint& k = F.mvKeysUn[i]; keys[i] = {k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave, k.class_id}; }
    for (auto& p : mps) { inV.push_back(p.mbTrackInView); bad.push_back(p.bad); obs.push_back(p.nobs > 0); px.push_back(p.mTrackProjX); py.push_back(p.mTrackProjY);
        pxr.push_back(p.mTrackProjXR); vc.push_back(p.mTrackViewCos); dep.push_back(p.mTrackDepth); lvl.push_back(p.mnTrackScaleLevel); md.insert(md.end(), p.desc.ptr(0), p.desc.ptr(0) + 32); }
    OFrame of = {F.N, keys.data(), F.mDescriptors.ptr(0), F.mvuRight.data(), occ.data(), F.mnMinX, F.mnMinY, F.mnMaxX, F.mnMaxY, F.mfGridElementWidthInv, F.mfGridElementHeightInv, F.mbf, 8, F.mvScaleFactors.data()};
    OMapPoints om = {(int)mps.size(), inV.data(), px.data(), py.data(), pxr.data(), lvl.data(), vc.data(), dep.data(), bad.data(), obs.data(), md.data()};
    std::vector<int> exp(F.N, -1);
    const int nexp = orbo_search_by_projection_mappoints(&of, &om, 3.0f, 1, 30.0f, 0.8f, exp.data());
    ORB_SLAM3::ORBmatcher matcher(0.8f);
    const int ngot = matcher.SearchByProjection(F, vp, 3.0f, true, 30.0f);
    int bad_assign = 0;
    for (int i = 0; i < F.N; i++) { MockMapPoint* e = exp[i] >= 0 ? vp[exp[i]] : nullptr; bad_assign += F.mvpMapPoints[i] != e; }
    const int dd = ORB_SLAM3::ORBmatcher::DescriptorDistance(F.mDescriptors.row(0), dR.row(0));
    // ---- ComputeStereoFishEyeMatches through the facade helper: a mock fisheye-rig Frame with the member names of include/Frame.h ----
    struct MockCam { float p[8]; float getParameter(int i) { return p[i]; } };
    struct M3 { float m[9]; float operator()(int r, int c) const { return m[3 * r + c]; } };
    struct V3 { float d[3]; V3() : d{0, 0, 0} {} V3(float a, float b, float c) : d{a, b, c} {} float operator()(int i) const { return d[i]; } };
    struct MockRigFrame {
        int Nleft = 0, Nright = 0, mnCloseMPs = 7; MockCam *mpCamera, *mpCamera2; M3 mRlr; V3 mtlr;
        std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch; std::vector<float> mvDepth, mvuRight; std::vector<V3> mvStereo3Dpoints;
    } RF;
    MockCam c1{{190.98f, 190.97f, 254.93f, 256.90f, 0.0034f, 0.0007f, -0.0020f, 0.0002f}}, c2{{190.44f, 190.43f, 252.60f, 254.92f, 0.0034f, 0.0018f, -0.0027f, 0.0003f}};
    RF.mpCamera = &c1; RF.mpCamera2 = &c2; RF.mRlr = M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; RF.mtlr = V3(0.101f, 0.0f, 0.0f);
    std::vector<cv::KeyPoint> kl2, kr2; cv::Mat dl2, dr2; std::vector<int> lap2 = {0, w - 1};
    exL(imL, cv::Mat(), kl2, dl2, lap2); exR(imR, cv::Mat(), kr2, dr2, lap2);
    RF.Nleft = (int)kl2.size(); RF.Nright = (int)kr2.size();
    ORB_SLAM3::ComputeStereoFishEyeMatches(RF, &exL, &exR);
    int nfish = 0, fish_bad = RF.mnCloseMPs != 0 || (int)RF.mvDepth.size() != RF.Nleft || (int)RF.mvRightToLeftMatch.size() != RF.Nright;
    for (int i = 0; i < RF.Nleft && !fish_bad; i++) {
        const int j = RF.mvLeftToRightMatch[i];
        if (j < 0) { fish_bad |= RF.mvDepth[i] != -1.0f; continue; }
        nfish++;
        fish_bad |= !(RF.mvDepth[i] > 0.0001f) || RF.mvStereo3Dpoints[i](2) != RF.mvDepth[i] || j >= RF.Nright || RF.mvRightToLeftMatch[j] < i;
    }
    printf("fisheye helper: %d of %d left keypoints matched, consistent=%d\n", nfish, RF.Nleft, !fish_bad);
    if (fish_bad) return 1;
    // ---- ComputeStereoFromRGBD through the facade helper, against the definition (src/Frame.cc:1361-1391) ----
    {
        MockFrame G;
        exL(imL, cv::Mat(), G.mvKeysUn, G.mDescriptors, lap);
        G.N = (int)G.mvKeysUn.size(); G.mbf = 40.0f;
        std::vector<float> depth((size_t)w * h);
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) depth[(size_t)y * w + x] = ((x * 7 + y * 3) % 11 == 0) ? 0.0f : 1.0f + 0.01f * (float)((x * 13 + y * 5) % 300);
        cv::Mat imD(h, w, CV_32F, depth.data());
        ORB_SLAM3::ComputeStereoFromRGBD(G, &exL, imD);
        int rgbd_bad = (int)G.mvuRight.size() != G.N || (int)G.mvDepth.size() != G.N, with_depth = 0;
        for (int i = 0; i < G.N && !rgbd_bad; i++) {
            const cv::KeyPoint& kp = G.mvKeysUn[i];
            const float d = depth[(size_t)(int)kp.pt.y * w + (int)kp.pt.x];
            const float eu = d > 0 ? kp.pt.x - G.mbf / d : -1.0f, ed = d > 0 ? d : -1.0f;
            rgbd_bad |= G.mvuRight[i] != eu || G.mvDepth[i] != ed; with_depth += d > 0;
        }
        printf("rgbd helper: %d of %d keypoints with depth, consistent=%d\n", with_depth, G.N, !rgbd_bad);
        if (rgbd_bad || with_depth < G.N / 2) return 1;
    }
    const int de = orbo_descriptor_distance(F.mDescriptors.ptr(0), dR.ptr(0));
    printf("N=%d stereo=%d matches facade=%d oracle=%d mismatched_assignments=%d dist %d %d TH %d %d %d\n", F.N, nstereo, ngot, nexp, bad_assign, dd, de,
           ORB_SLAM3::ORBmatcher::TH_LOW, ORB_SLAM3::ORBmatcher::TH_HIGH, ORB_SLAM3::ORBmatcher::HISTO_LENGTH);
    return (ngot == nexp && bad_assign == 0 && dd == de && nstereo > 20 && ngot > 50) ? 0 : 1;
}
