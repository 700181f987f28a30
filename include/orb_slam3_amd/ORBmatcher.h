/* ORBmatcher.h — drop-in facade for the part of ORB_SLAM3::ORBmatcher that this library accelerates
 * (/root/reference/include/ORBmatcher.h:36-103): constructor, DescriptorDistance, SearchByProjection(Frame&, vector<MapPoint*>&,...)
 * and the public TH_LOW / TH_HIGH / HISTO_LENGTH constants, plus ComputeStereoMatches (the body of Frame::ComputeStereoMatches,
 * src/Frame.cc:1102-1358, as a function of the Frame).
 *
 * The methods are templates on the Frame / MapPoint types so that this header compiles both against the reference's real
 * classes (member names below are the reference's, include/Frame.h and include/MapPoint.h) and against the light mock structs of
 * tests/cpp/matcher_facade_test.cpp — the reference's Frame.h itself needs Eigen/Sophus/DBoW2, which are not available in the
 * build container.  The remaining overloads (Frame/Frame projection, SearchForTriangulation) are bound the same way from the C ABI
 * (orbm_search_by_projection_frame, orbm_search_for_triangulation); INTEGRATION.md has the snippets, since their view construction
 * needs Sophus poses.
 */
#ifndef ORB_SLAM3_AMD_ORBMATCHER_H
#define ORB_SLAM3_AMD_ORBMATCHER_H

#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>
#include <opencv2/opencv.hpp>
#include "../orbx.h"

namespace ORB_SLAM3
{

class ORBmatcher
{
public:
    ORBmatcher(float nnratio=0.6, bool checkOri=true): mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // Computes the Hamming distance between two ORB descriptors (src/ORBmatcher.cc:2383).  One pair per call is a GPU round
    // trip: hot N^2 callers should use DescriptorDistances.
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
    {
        int d = 0;
        std::lock_guard<std::mutex> lock(Mutex());
        Check(orbm_hamming_matrix(SharedHandle(), a.ptr(0), 1, b.ptr(0), 1, &d));
        return d;
    }
    // all pairs: out[i*B.rows + j] = distance(A.row(i), B.row(j))
    static void DescriptorDistances(const cv::Mat &A, const cv::Mat &B, std::vector<int>& out)
    {
        out.resize((size_t)A.rows * B.rows);
        std::lock_guard<std::mutex> lock(Mutex());
        if (A.rows && B.rows) Check(orbm_hamming_matrix(SharedHandle(), A.ptr(0), A.rows, B.ptr(0), B.rows, out.data()));
    }

    // Search matches between Frame keypoints and projected MapPoints. Returns number of matches
    // Used to track the local map (Tracking)   — src/ORBmatcher.cc:45-239, non-fisheye path (F.Nleft == -1)
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT &F, const std::vector<MapPointT*> &vpMapPoints, const float th=3, const bool bFarPoints = false, const float thFarPoints = 50.0f)
    {
        if (F.Nleft != -1) throw std::runtime_error("ORBmatcher (HIP): the fisheye (Nleft != -1) path is not accelerated");
        const int N = F.N, M = (int)vpMapPoints.size();
        std::vector<OrbxKeyPoint> keys(N); std::vector<uint8_t> occ(N, 0);
        for (int i = 0; i < N; i++) {
            const cv::KeyPoint& k = F.mvKeysUn[i];
            keys[i].x = k.pt.x; keys[i].y = k.pt.y; keys[i].size = k.size; keys[i].angle = k.angle; keys[i].response = k.response; keys[i].octave = k.octave; keys[i].class_id = k.class_id;
            if (F.mvpMapPoints[i]) occ[i] = F.mvpMapPoints[i]->Observations() > 0;
        }
        std::vector<uint8_t> inView(M), bad(M), hasObs(M), desc((size_t)M * 32);
        std::vector<float> px(M), py(M), pxr(M), vcos(M), depth(M); std::vector<int> lvl(M);
        for (int i = 0; i < M; i++) {
            MapPointT* p = vpMapPoints[i];
            inView[i] = p->mbTrackInView; bad[i] = p->isBad(); hasObs[i] = p->Observations() > 0;
            px[i] = p->mTrackProjX; py[i] = p->mTrackProjY; pxr[i] = p->mTrackProjXR; vcos[i] = p->mTrackViewCos; depth[i] = p->mTrackDepth;
            lvl[i] = p->mnTrackScaleLevel;
            const cv::Mat d = p->GetDescriptor();
            memcpy(&desc[(size_t)i * 32], d.ptr(0), 32);
        }
        OrbmFrameView fv; FillFrameView(F, keys, occ, fv);
        OrbmMapPointView mv = {M, inView.data(), px.data(), py.data(), pxr.data(), lvl.data(), vcos.data(), depth.data(), bad.data(), hasObs.data(), desc.data()};
        std::vector<int> assigned(N > 0 ? N : 1, -1);
        int nmatches = 0;
        std::lock_guard<std::mutex> lock(Mutex());      // the reference's matchers are re-entrant (Tracking / LocalMapping / LoopClosing threads); one handle is not
        Check(orbm_search_by_projection_mappoints(SharedHandle(), &fv, &mv, th, bFarPoints, thFarPoints, mfNNratio, assigned.data(), &nmatches));
        for (int i = 0; i < N; i++) if (assigned[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[assigned[i]];
        return nmatches;
    }

public:
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

    static orbx_extractor* SharedHandle()
    {
        static orbx_extractor* h = nullptr;
        static std::once_flag once;
        std::call_once(once, [](){ if (orbx_create(&h, 1000, 1.2f, 8, 20, 7, 0) != ORBX_OK) h = nullptr; });
        if (!h) throw std::runtime_error(std::string("ORBmatcher (HIP): ") + orbx_last_error());
        return h;
    }
    static std::mutex& Mutex() { static std::mutex m; return m; }
    static void Check(int rc) { if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher (HIP): ") + orbx_last_error()); }

protected:
    template <class FrameT>
    static void FillFrameView(FrameT& F, const std::vector<OrbxKeyPoint>& keys, const std::vector<uint8_t>& occ, OrbmFrameView& fv)
    {
        fv.N = F.N; fv.keys_un = keys.data(); fv.desc = F.mDescriptors.ptr(0); fv.u_right = F.mvuRight.empty() ? nullptr : F.mvuRight.data();
        fv.occupied = occ.data(); fv.min_x = F.mnMinX; fv.min_y = F.mnMinY; fv.max_x = F.mnMaxX; fv.max_y = F.mnMaxY;
        fv.grid_w_inv = F.mfGridElementWidthInv; fv.grid_h_inv = F.mfGridElementHeightInv; fv.mbf = F.mbf;
        fv.nlevels = (int)F.mvScaleFactors.size(); fv.scale_factors = F.mvScaleFactors.data();
    }

    float mfNNratio;
    bool mbCheckOrientation;
};

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

// Frame::ComputeStereoMatches (src/Frame.cc:1102-1358) for a Frame whose two extractors are the HIP facades: runs on the
// device-resident pyramids/descriptors of the LAST call of each extractor and fills F.mvuRight / F.mvDepth.
template <class FrameT, class ExtractorT>
void ComputeStereoMatches(FrameT& F, ExtractorT* pLeft, ExtractorT* pRight)
{
    const int cap = orbx_max_keypoints(pLeft->Handle());
    std::vector<float> u(cap, -1.0f), d(cap, -1.0f);
    int n = 0;
    ORBmatcher::Check(orbm_stereo_match(pLeft->Handle(), 0, pRight->Handle(), 0, 1, F.mbf, F.mb));
    ORBmatcher::Check(orbm_stereo_fetch(pLeft->Handle(), 1, u.data(), d.data(), cap, &n));
    F.mvuRight.assign(u.begin(), u.begin() + F.N);
    F.mvDepth.assign(d.begin(), d.begin() + F.N);
}

} // namespace ORB_SLAM

#endif // ORB_SLAM3_AMD_ORBMATCHER_H
