/* ORBmatcher.h — drop-in facade for ORB_SLAM3::ORBmatcher (/root/reference/include/ORBmatcher.h:36-103): every public method of the
 * reference class, with the reference's signatures, on top of the C ABI of include/orbx.h, plus ComputeStereoMatches (the body of
 * Frame::ComputeStereoMatches, src/Frame.cc:1102-1358, as a function of the Frame).
 *
 * Division of labour (the same for every method): the facade reads the raw fields the method needs through the caller's own accessors
 * (GetWorldPos, GetNormal, the distance limits, descriptors, poses, camera parameters) and applies the method's object-level tests (bad,
 * already matched, ...); the geometry in front of the window search - pose * point, projection, image test, distance and viewing-angle gates -
 * runs on the device for all points at once (orbm_project_points, PointBatch below), as do the window search, the level gates, every
 * Hamming distance and the accept loop; MapPoint::PredictScale is asked of the caller's MapPoint for the survivors (it reads a protected
 * member); the facade then writes the result back into the caller's objects exactly where the reference would have.
 *
 * The methods are templates on the Frame / KeyFrame / MapPoint (and Sim3) types, and the geometric types are taken from the return types
 * of their accessors, so this header needs neither Eigen nor Sophus itself.  It compiles against the reference's real classes (member
 * names below are the reference's: include/Frame.h, include/KeyFrame.h, include/MapPoint.h), against the stand-in world of
 * oracle/slam_shim (tests/cpp/matcher_world_driver.cpp drives this facade and the reference's own compiled ORBmatcher.cc with identical
 * objects and compares every output) and against the light mocks of tests/cpp/matcher_facade_test.cpp.
 *
 * The fisheye branches of SearchForTriangulation (KannalaBrandt8::epipolarConstrain = the triangulation test) run on the device as well;
 * the 4x4 null vector of the triangulation is an fp64 eigen-decomposition there instead of Eigen's fp32 JacobiSVD (see kb8_model.h).
 */
#ifndef ORB_SLAM3_AMD_ORBMATCHER_H
#define ORB_SLAM3_AMD_ORBMATCHER_H
// This header REPLACES the reference's include/ORBmatcher.h and takes its include guard (see ORBextractor.h)
#ifdef ORBMATCHER_H
#error "the reference's include/ORBmatcher.h was included before the drop-in ORBmatcher.h: replace that file with this one, or put this directory first on the include path; see INTEGRATION.md section 4"
#endif
#define ORBMATCHER_H

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <atomic>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>
#include <opencv2/opencv.hpp>
#include "../orbx.h"

namespace ORB_SLAM3
{

// TH_LOW / TH_HIGH / HISTO_LENGTH (src/ORBmatcher.cc:35-37) as static members of a class template: their definitions may then live in this
// header and still be merged to one object when several translation units (Frame.cc, MapPoint.cc, Tracking.cc, ...) include it.
template <class Unused> struct ORBmatcherConstants { static const int TH_LOW; static const int TH_HIGH; static const int HISTO_LENGTH; };
template <class Unused> const int ORBmatcherConstants<Unused>::TH_HIGH = 100;
template <class Unused> const int ORBmatcherConstants<Unused>::TH_LOW = 50;
template <class Unused> const int ORBmatcherConstants<Unused>::HISTO_LENGTH = 30;

class ORBmatcher : public ORBmatcherConstants<void>
{
    template <class X> using Decay = typename std::decay<X>::type;

public:
    ORBmatcher(float nnratio=0.6, bool checkOri=true): mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // Device-resident key frames (orbm_keyframe, include/orbx.h): what the vocabulary-bucket searches read of a key frame - mvKeysUn,
    // mDescriptors, mvuRight, mFeatVec - is fixed once ComputeBoW has run, so it is uploaded on first use and kept until Erase (call it where
    // the map erases the key frame, src/Map.cc:106-124) or the end of the cache.  One cache per thread that owns key frames' searches
    // (LocalMapping, LoopClosing); the device objects can be used from any thread's handle.
    template <class KeyFrameT>
    class ResidentKeyFrames
    {
    public:
        ResidentKeyFrames() {}
        ~ResidentKeyFrames() { Clear(); }
        // The cache is keyed by address; an entry also remembers which key frame it was made from (mnId, N, the size of mFeatVec) and is
        // rebuilt when that no longer fits - an address reused by a new key frame, or ComputeBoW having run since.  A key frame whose
        // mFeatVec is still empty (ComputeBoW has not run) is uploaded for this call only and not cached.
        orbm_keyframe* Get(KeyFrameT* pKF)
        {
            // device objects belong to one GPU: when SetDevice / SetThreadDevice has moved this thread to another one, everything cached for the
            // previous GPU is dropped and uploaded again on demand (a search would otherwise refuse the pair: "bad key frame")
            if (device != Device()) { Clear(); device = Device(); }
            const Tag tag = {(long long)pKF->mnId, (long long)pKF->N, (long long)pKF->mFeatVec.size(), Fingerprint(pKF)};
            auto it = m.find(pKF);
            if (it != m.end()) {
                if (it->second.tag == tag) { it->second.stamp = ++clock; return it->second.dev; }
                orbm_keyframe_destroy(it->second.dev); m.erase(it);
            }
            if (transient) { orbm_keyframe_destroy(transient); transient = nullptr; }
            BowStore k; FillBow(*pKF, pKF->N, k);
            if (pKF->NLeft != -1) FillAllKeys(*pKF, pKF->NLeft, pKF->N, k);      // a rig: mvKeys followed by mvKeysRight (:1136-1141)
            if (pKF->mpCamera2) k.v.u_right = nullptr;                            // bStereo = !mpCamera2 && mvuRight[idx] >= 0 (:1126)
            orbm_keyframe* r = nullptr;
            {
                Check(orbm_keyframe_create(SharedHandle(), &k.v, &r));
            }
            if (pKF->N > 0 && pKF->mFeatVec.empty()) { transient = r; return r; }
            Entry e; e.dev = r; e.tag = tag; e.stamp = ++clock;
            m[pKF] = e;
            return r;
        }
        void Erase(KeyFrameT* pKF) { auto it = m.find(pKF); if (it != m.end()) { orbm_keyframe_destroy(it->second.dev); m.erase(it); } }
        void Clear() { for (auto& e : m) orbm_keyframe_destroy(e.second.dev); m.clear(); if (transient) { orbm_keyframe_destroy(transient); transient = nullptr; } }
        size_t size() const { return m.size(); }
        // Bound on the number of resident key frames (0 = none, the default of an explicit cache whose owner calls Erase).  Trim() - called by the
        // searches BEFORE they collect the key frames of a call, never while device objects of the call are in use - drops the least recently
        // used entries down to the bound; one call may exceed it by the key frames it names.
        void SetCapacity(size_t n) { capacity = n; }
        size_t Capacity() const { return capacity; }
        void Trim()
        {
            if (capacity == 0 || m.size() <= capacity) return;
            std::vector<std::pair<unsigned long long, KeyFrameT*> > byAge;
            byAge.reserve(m.size());
            for (auto& e : m) byAge.push_back(std::make_pair(e.second.stamp, e.first));
            const size_t drop = m.size() - capacity + capacity / 8;            // a little below the bound: not one eviction per insertion
            std::nth_element(byAge.begin(), byAge.begin() + (drop < byAge.size() ? drop : byAge.size() - 1), byAge.end());
            for (size_t i = 0; i < drop && i < byAge.size(); i++) Erase(byAge[i].second);
        }
    private:
        ResidentKeyFrames(const ResidentKeyFrames&);
        ResidentKeyFrames& operator=(const ResidentKeyFrames&);
        struct Tag { long long id, n, nodes; unsigned long long print; bool operator==(const Tag& o) const { return id == o.id && n == o.n && nodes == o.nodes && print == o.print; } };
        // eight descriptor rows spread over the key frame, folded into 64 bits: tells apart two key frames that reuse an address AND an id (a second SLAM
        // system in the process, a test that builds world after world); 256 bytes read per lookup.  Like every search of the reference it reads
        // mDescriptors without a lock: the matrix is written once, by the Frame the key frame is made from, before the key frame is published.
        static unsigned long long Fingerprint(KeyFrameT* pKF)
        {
            unsigned long long f = 0x9E3779B97F4A7C15ull;
            const int n = pKF->mDescriptors.rows;
            for (int k = 0; k < 8 && n > 0; k++) {
                unsigned long long w[4];
                memcpy(w, pKF->mDescriptors.ptr((int)((long long)k * (n - 1) / 7)), 32);
                for (int i = 0; i < 4; i++) { f ^= w[i]; f *= 0x100000001B3ull; f ^= f >> 29; }
            }
            return f;
        }
        struct Entry { orbm_keyframe* dev; Tag tag; unsigned long long stamp; };
        std::map<KeyFrameT*, Entry> m;
        orbm_keyframe* transient = nullptr;
        int device = -1;
        size_t capacity = 0;
        unsigned long long clock = 0;
    };
    // The cache behind the reference's own single-call signatures (SearchByBoW(pKF, F, ...) of Tracking.cc:3183 / :4371, SearchForTriangulation(pKF1, pKF2, ...)
    // of LocalMapping.cc:610): one per calling thread and key-frame type, so an unchanged call site pays the upload of a key frame once, not per call.
    // An unchanged caller has no Erase to call when the map drops a key frame, so this cache is bounded: at most kImplicitCacheEntries key frames
    // (about the covisibility window LocalMapping and Tracking revisit) stay resident per thread, least recently used first out - ~130 KB each at
    // 2 000 keypoints (64 B per keypoint + the FeatureVector), i.e. <= 33 MB of device memory per thread and key-frame type.  A host that edits
    // Map::EraseKeyFrame anyway can release an entry at once: ORBmatcher::EraseImplicit(pKF) on the thread that searched it (or on every thread:
    // an entry of a deleted key frame is never matched again - the tag of a new key frame at the same address differs - and ages out).
    enum { kImplicitCacheEntries = 256 };
    template <class KeyFrameT>
    static ResidentKeyFrames<KeyFrameT>& ImplicitCache()
    {
        static thread_local ResidentKeyFrames<KeyFrameT> c;
        if (c.Capacity() == 0) c.SetCapacity((size_t)kImplicitCacheEntries);
        return c;
    }
    template <class KeyFrameT> static void EraseImplicit(KeyFrameT* pKF) { ImplicitCache<KeyFrameT>().Erase(pKF); }
    template <class KeyFrameT> static void ClearImplicit() { ImplicitCache<KeyFrameT>().Clear(); }
    template <class KeyFrameT> static void SetImplicitCapacity(size_t n) { ImplicitCache<KeyFrameT>().SetCapacity(n ? n : 1); }

    // Computes the Hamming distance between two ORB descriptors (src/ORBmatcher.cc:2383).
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
    {
        // one pair = four 64-bit popcounts: done where the caller is (MapPoint::ComputeDistinctiveDescriptors calls this N^2 times,
        // src/MapPoint.cc:496; Frame::ComputeStereoMatches once per candidate, src/Frame.cc:1219); the batched forms run on the device
        unsigned long long x[4], y[4];
        memcpy(x, a.ptr(0), 32); memcpy(y, b.ptr(0), 32);
        return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
    }
    // all pairs: out[i*B.rows + j] = distance(A.row(i), B.row(j))
    static void DescriptorDistances(const cv::Mat &A, const cv::Mat &B, std::vector<int>& out)
    {
        out.resize((size_t)A.rows * B.rows);
        if (A.rows && B.rows) Check(orbm_hamming_matrix(SharedHandle(), A.ptr(0), A.rows, B.ptr(0), B.rows, out.data()));
    }

    // Search matches between Frame keypoints and projected MapPoints. Returns number of matches
    // Used to track the local map (Tracking)   — src/ORBmatcher.cc:45-239, both the one-camera and the fisheye-rig (F.Nleft != -1) path
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT &F, const std::vector<MapPointT*> &vpMapPoints, const float th=3, const bool bFarPoints = false, const float thFarPoints = 50.0f)
    {
        const int M = (int)vpMapPoints.size();
        const bool rig = F.Nleft != -1;
        std::vector<uint8_t> inView(M), bad(M), hasObs(M), desc((size_t)M * 32), inViewR(M);
        std::vector<float> px(M), py(M), pxr(M), vcos(M), depth(M), pyr(M), vcosR(M); std::vector<int> lvl(M), lvlR(M);
        for (int i = 0; i < M; i++) {
            MapPointT* p = vpMapPoints[i];
            inView[i] = p->mbTrackInView; inViewR[i] = p->mbTrackInViewR; bad[i] = p->isBad(); hasObs[i] = p->Observations() > 0;
            px[i] = p->mTrackProjX; py[i] = p->mTrackProjY; pxr[i] = p->mTrackProjXR; pyr[i] = p->mTrackProjYR; vcos[i] = p->mTrackViewCos; vcosR[i] = p->mTrackViewCosR;
            depth[i] = p->mTrackDepth; lvl[i] = p->mnTrackScaleLevel; lvlR[i] = p->mnTrackScaleLevelR;
            CopyDescriptor(p, &desc[(size_t)i * 32]);
        }
        OrbmMapPointView mv = {M, inView.data(), px.data(), py.data(), pxr.data(), lvl.data(), vcos.data(), depth.data(), bad.data(), hasObs.data(), desc.data()};
        int nmatches = 0;
        std::vector<int> assigned;
        if (!rig) {
            FrameStore fs; FillFrame(F, fs, OccupiedWithObservations);
            assigned.assign(F.N > 0 ? F.N : 1, -1);
            Check(orbm_search_by_projection_mappoints(SharedHandle(), &fs.v, &mv, th, bFarPoints, thFarPoints, mfNNratio, assigned.data(), &nmatches));
        } else {
            RigStore rs; FillRig(F, rs, OccupiedWithObservations);
            OrbmMapPointRightView mr = {inViewR.data(), pxr.data(), pyr.data(), lvlR.data(), vcosR.data()};
            assigned.assign(F.N > 0 ? F.N : 1, -1);
            Check(orbm_search_by_projection_mappoints_fisheye(SharedHandle(), &rs.v, &mv, &mr, th, bFarPoints, thFarPoints, mfNNratio, assigned.data(), &nmatches));
        }
        for (int i = 0; i < F.N; i++) if (assigned[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[assigned[i]];
        return nmatches;
    }

    // Project MapPoints tracked in last frame into the current frame and search matches.
    // Used to track from previous frame (Tracking)   — src/ORBmatcher.cc:1950-2184
    template <class FrameT>
    int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono)
    {
        // motion along the optical axis between the two frames picks the pyramid levels to search (:1973-1975)
        const auto poseCur = CurrentFrame.GetPose();
        const float zLastOfCurCentre = (LastFrame.GetPose() * poseCur.inverse().translation())(2);
        const bool towards = !bMono && zLastOfCurCentre > CurrentFrame.mb, away = !bMono && -zLastOfCurCentre > CurrentFrame.mb;
        const bool twoCameras = CurrentFrame.Nleft != -1;

        const int NL = LastFrame.N;
        PointBatch pts(NL);
        std::vector<uint8_t> seen(NL > 0 ? NL : 1, 0), desc((size_t)(NL > 0 ? NL : 1) * 32, 0);
        std::vector<float> ang(NL > 0 ? NL : 1, 0); std::vector<int> oct(NL > 0 ? NL : 1, 0);
        for (int i = 0; i < NL; i++) {
            auto* pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;
            pts.take(i, pMP, false);
            const bool cam1 = LastFrame.Nleft == -1 || i < LastFrame.Nleft;
            oct[i] = cam1 ? LastFrame.mvKeys[i].octave : LastFrame.mvKeysRight[i - LastFrame.Nleft].octave;
            ang[i] = (LastFrame.Nleft == -1 ? LastFrame.mvKeysUn[i] : cam1 ? LastFrame.mvKeys[i] : LastFrame.mvKeysRight[i - LastFrame.Nleft]).angle;
            seen[i] = pMP->Observations() > 0;
            CopyDescriptor(pMP, &desc[(size_t)i * 32]);
        }
        OrbmProjection spec = Spec(poseCur, CurrentFrame.mpCamera, CurrentFrame, /*frameBounds*/ 0);
        spec.depth_test = 2;                                  // 1 / z < 0 rejects (:1997-2000)
        pts.project(spec);
        std::vector<float> uR, vR;
        if (twoCameras) {                                     // the same points through the left-to-right transform, camera-1 model (:2092-2094)
            OrbmProjection right = spec;
            right.depth_test = 0; right.bounds_mode = 2;
            SpecSecond(right, CurrentFrame.GetRelativePoseTrl());
            PointBatch again = pts;
            again.project(right);
            uR.swap(again.u); vR.swap(again.v);
        }
        OrbmLastFrameView lv = {NL, pts.valid.data(), pts.u.data(), pts.v.data(), pts.invz.data(), oct.data(), ang.data(), seen.data(), desc.data()};
        std::vector<int> assigned(CurrentFrame.N > 0 ? CurrentFrame.N : 1, -1);
        int nmatches = 0;
        if (!twoCameras) {
            FrameStore fs; FillFrame(CurrentFrame, fs, OccupiedWithObservations);
            Check(orbm_search_by_projection_frame(SharedHandle(), &fs.v, &lv, th, towards, away, mbCheckOrientation, assigned.data(), &nmatches));
        } else {
            RigStore rs; FillRig(CurrentFrame, rs, OccupiedWithObservations);
            Check(orbm_search_by_projection_frame_fisheye(SharedHandle(), &rs.v, &lv, uR.data(), vR.data(), th, towards, away, mbCheckOrientation,
                                                          assigned.data(), &nmatches));
        }
        for (int i = 0; i < CurrentFrame.N; i++) {
            if (assigned[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[assigned[i]];
            else if (assigned[i] == -2) CurrentFrame.mvpMapPoints[i] = nullptr;
        }
        return nmatches;
    }

    // Project MapPoints seen in KeyFrame into the Frame and search matches.
    // Used in relocalisation (Tracking)   — src/ORBmatcher.cc:2196-2324
    template <class FrameT, class KeyFrameT, class MapPointT>
    int SearchByProjection(FrameT &CurrentFrame, KeyFrameT* pKF, const std::set<MapPointT*> &sAlreadyFound, const float th, const int ORBdist)
    {
        const auto pose = CurrentFrame.GetPose();
        const std::vector<MapPointT*> kfPoints = pKF->GetMapPointMatches();
        const int M = (int)kfPoints.size();
        PointBatch pts(M);
        for (int i = 0; i < M; i++) {
            MapPointT* pMP = kfPoints[i];
            if (pMP && !pMP->isBad() && !sAlreadyFound.count(pMP)) pts.take(i, pMP, false);
        }
        OrbmProjection spec = Spec(pose, CurrentFrame.mpCamera, CurrentFrame, /*frameBounds*/ 0);
        SpecCentre(spec, pose.inverse().translation());
        spec.distance_test = 1;                               // no depth test, no viewing angle in this method (:2228-2256)
        pts.project(spec);
        ProjStore ps(M);
        for (int i = 0; i < M; i++) {
            if (!pts.valid[i]) continue;
            ps.set(i, pts.u[i], pts.v[i], 0.0f, kfPoints[i]->PredictScale(pts.dist[i], &CurrentFrame), pKF->mvKeysUn[i].angle);
            CopyDescriptor(kfPoints[i], ps.descAt(i));
        }
        FrameStore fs; FillFrame(CurrentFrame, fs, OccupiedAny);
        std::vector<int> assigned(CurrentFrame.N > 0 ? CurrentFrame.N : 1, -1);
        int nmatches = 0;
        {
            Check(orbm_search_by_projection_keyframe(SharedHandle(), &fs.v, ps.view(), th, ORBdist, mbCheckOrientation, assigned.data(), &nmatches));
        }
        for (int i = 0; i < CurrentFrame.N; i++) {
            if (assigned[i] >= 0) CurrentFrame.mvpMapPoints[i] = kfPoints[assigned[i]];
            else if (assigned[i] == -2) CurrentFrame.mvpMapPoints[i] = nullptr;
        }
        return nmatches;
    }

    // Project MapPoints using a Similarity Transformation and search matches.
    // Used in loop detection (Loop Closing)   — src/ORBmatcher.cc:495-606
    template <class KeyFrameT, class Sim3T, class MapPointT>
    int SearchByProjection(KeyFrameT* pKF, Sim3T &Scw, const std::vector<MapPointT*> &vpPoints, std::vector<MapPointT*> &vpMatched, int th, float ratioHamming=1.0)
    {
        std::vector<int> assigned;
        const int n = SearchBySim3Projection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming, /*inlineProjection=*/false, assigned);
        for (size_t idx = 0; idx < vpMatched.size(); idx++) if (assigned[idx] >= 0) vpMatched[idx] = vpPoints[assigned[idx]];
        return n;
    }

    // Project MapPoints using a Similarity Transformation and search matches.
    // Used in Place Recognition (Loop Closing and Merging)   — src/ORBmatcher.cc:608-732
    template <class KeyFrameT, class Sim3T, class MapPointT>
    int SearchByProjection(KeyFrameT* pKF, Sim3T &Scw, const std::vector<MapPointT*> &vpPoints, const std::vector<KeyFrameT*> &vpPointsKFs,
                           std::vector<MapPointT*> &vpMatched, std::vector<KeyFrameT*> &vpMatchedKF, int th, float ratioHamming=1.0)
    {
        std::vector<int> assigned;
        const int n = SearchBySim3Projection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming, /*inlineProjection=*/true, assigned);
        for (size_t idx = 0; idx < vpMatched.size(); idx++)
            if (assigned[idx] >= 0) { vpMatched[idx] = vpPoints[assigned[idx]]; vpMatchedKF[idx] = vpPointsKFs[assigned[idx]]; }
        return n;
    }

    // Search matches between MapPoints in a KeyFrame and ORB in a Frame.
    // Brute force constrained to ORB that belong to the same vocabulary node (at a certain level)
    // Used in Relocalisation and Loop Detection   — src/ORBmatcher.cc:259-493 (one camera)
    template <class KeyFrameT, class FrameT, class MapPointT>
    int SearchByBoW(KeyFrameT* pKF, FrameT &F, std::vector<MapPointT*> &vpMapPointMatches)
    {
        if (F.Nleft == -1 && !pKF->mpCamera2 && !pKF->mFeatVec.empty()) {
            // one camera: the key frame stays on the device between calls (ImplicitCache), the frame goes up once, the accept loop runs on the device
            std::vector<std::vector<MapPointT*> > vv;
            const std::vector<int> counts = SearchByBoW(std::vector<KeyFrameT*>(1, pKF), F, ImplicitCache<KeyFrameT>(), vv);
            vpMapPointMatches.swap(vv[0]);
            return counts[0];
        }
        const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPointT*>(F.N,static_cast<MapPointT*>(NULL));
        const bool rig = F.Nleft != -1;
        BowStore k1, k2;
        if (!rig && !pKF->mpCamera2) { FillBow(*pKF, pKF->N, k1); FillBow(F, F.N, k2); }
        else {                                              // keypoints by feature index: camera 1 (mvKeys), then camera 2 (mvKeysRight) (:392-396, :399-402)
            FillBow(*pKF, pKF->N, k1); FillBow(F, F.N, k2);
            FillAllKeys(*pKF, pKF->NLeft, pKF->N, k1); FillAllKeys(F, F.Nleft, F.N, k2);
        }
        k1.present.assign(pKF->N > 0 ? pKF->N : 1, 0);
        for (int i = 0; i < pKF->N; i++) k1.present[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
        k1.v.has_map_point = k1.present.data();
        k2.v.has_map_point = nullptr;
        int nmatches = 0;
        if (!rig) {
            std::vector<int> m12(pKF->N > 0 ? pKF->N : 1, -1);
            {
                Check(orbm_search_by_bow(SharedHandle(), &k1.v, &k2.v, mfNNratio, 1, mbCheckOrientation, m12.data(), &nmatches));
            }
            for (int i = 0; i < pKF->N; i++) if (m12[i] >= 0) vpMapPointMatches[m12[i]] = vpMapPointsKF[i];
        } else {
            std::vector<int> a2(F.N > 0 ? F.N : 1, -1);
            {
                Check(orbm_search_by_bow_fisheye(SharedHandle(), &k1.v, &k2.v, F.Nleft, mfNNratio, mbCheckOrientation, a2.data(), &nmatches));
            }
            for (int j = 0; j < F.N; j++) if (a2[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[a2[j]];
        }
        return nmatches;
    }
    // The same for several candidate key frames against one frame in one call over device-resident key frames (Tracking::Relocalization
    // calls SearchByBoW(pKF, mCurrentFrame, vvpMapPointMatches[i]) for every candidate, src/Tracking.cc:4360-4380): the frame is uploaded once
    // for all candidates, the key frames come from `cache`, the accept loop runs on the device.  One camera (F.Nleft == -1).
    // Returns the match counts; vvpMapPointMatches[i] is what SearchByBoW(vpKFs[i], F, ...) gives.
    template <class KeyFrameT, class FrameT, class MapPointT>
    std::vector<int> SearchByBoW(const std::vector<KeyFrameT*>& vpKFs, FrameT &F, ResidentKeyFrames<KeyFrameT>& cache,
                                 std::vector<std::vector<MapPointT*> > &vvpMapPointMatches)
    {
        const int n = (int)vpKFs.size();
        vvpMapPointMatches.assign(n, std::vector<MapPointT*>(F.N, static_cast<MapPointT*>(NULL)));
        std::vector<int> counts(n, 0);
        if (n == 0) return counts;
        if (F.Nleft != -1) throw std::runtime_error("ORBmatcher (HIP): the batched SearchByBoW covers frames of one camera");
        cache.Trim();
        BowStore kf; FillBow(F, F.N, kf);
        orbm_keyframe* rf = nullptr;
        std::vector<orbm_keyframe*> k1(n), k2(n);
        std::vector<std::vector<MapPointT*> > mps(n);
        std::vector<std::vector<uint8_t> > good(n);
        std::vector<const uint8_t*> goodp(n);
        std::vector<std::vector<int> > m12(n);
        std::vector<int*> m12p(n);
        for (int i = 0; i < n; i++) {
            KeyFrameT* pKF = vpKFs[i];
            if (pKF->mpCamera2) throw std::runtime_error("ORBmatcher (HIP): the batched SearchByBoW covers key frames of one camera");
            k1[i] = cache.Get(pKF);
            mps[i] = pKF->GetMapPointMatches();
            good[i].assign(pKF->N > 0 ? pKF->N : 1, 0);
            for (int j = 0; j < pKF->N; j++) good[i][j] = mps[i][j] && !mps[i][j]->isBad();
            goodp[i] = good[i].data();
            m12[i].assign(pKF->N > 0 ? pKF->N : 1, -1); m12p[i] = m12[i].data();
        }
        {
            Check(orbm_keyframe_create(SharedHandle(), &kf.v, &rf));
            for (int i = 0; i < n; i++) k2[i] = rf;
            const int rc = orbm_search_by_bow_resident(SharedHandle(), n, k1.data(), goodp.data(), k2.data(), nullptr, mfNNratio, 1, mbCheckOrientation, m12p.data(), counts.data());
            orbm_keyframe_destroy(rf);
            Check(rc);
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < vpKFs[i]->N; j++) if (m12[i][j] >= 0) vvpMapPointMatches[i][m12[i][j]] = mps[i][j];
        return counts;
    }
    // src/ORBmatcher.cc:892-1043; on fisheye-rig key frames only the features of camera 1 take part (:930, :953)
    template <class KeyFrameT, class MapPointT>
    int SearchByBoW(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*> &vpMatches12)
    {
        const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches();
        const std::vector<MapPointT*> vpMapPoints2 = pKF2->GetMapPointMatches();
        vpMatches12 = std::vector<MapPointT*>(vpMapPoints1.size(),static_cast<MapPointT*>(NULL));
        BowStore k1, k2;
        FillBow(*pKF1, (int)vpMapPoints1.size(), k1); FillBow(*pKF2, (int)vpMapPoints2.size(), k2);
        k1.present.assign(vpMapPoints1.size() > 0 ? vpMapPoints1.size() : 1, 0); k2.present.assign(vpMapPoints2.size() > 0 ? vpMapPoints2.size() : 1, 0);
        const size_t lim1 = pKF1->NLeft != -1 ? pKF1->mvKeysUn.size() : vpMapPoints1.size(), lim2 = pKF2->NLeft != -1 ? pKF2->mvKeysUn.size() : vpMapPoints2.size();
        for (size_t i = 0; i < vpMapPoints1.size() && i < lim1; i++) k1.present[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();
        for (size_t i = 0; i < vpMapPoints2.size() && i < lim2; i++) k2.present[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();
        k1.v.has_map_point = k1.present.data(); k2.v.has_map_point = k2.present.data();
        std::vector<int> m12(vpMapPoints1.size() > 0 ? vpMapPoints1.size() : 1, -1);
        int nmatches = 0;
        {
            Check(orbm_search_by_bow(SharedHandle(), &k1.v, &k2.v, mfNNratio, 0, mbCheckOrientation, m12.data(), &nmatches));
        }
        for (size_t i = 0; i < vpMapPoints1.size(); i++) if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
        return nmatches;
    }

    // Matching for the Map Initialization (only used in the monocular case)   — src/ORBmatcher.cc:734-880
    template <class FrameT>
    int SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize=10)
    {
        const int N1 = (int)F1.mvKeysUn.size();
        vnMatches12 = std::vector<int>(N1,-1);
        FrameStore f1, f2; FillFrame(F1, f1); FillFrame(F2, f2);
        std::vector<float> prev((size_t)N1 * 2 + 2);
        for (int i = 0; i < N1; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
        int nmatches = 0;
        std::vector<int> m12(N1 > 0 ? N1 : 1, -1);
        {
            Check(orbm_search_for_initialization(SharedHandle(), &f1.v, &f2.v, prev.data(), windowSize, mfNNratio, mbCheckOrientation, m12.data(), &nmatches));
        }
        for (int i = 0; i < N1; i++) { vnMatches12[i] = m12[i]; vbPrevMatched[i].x = prev[2 * i]; vbPrevMatched[i].y = prev[2 * i + 1]; }
        return nmatches;
    }

    // epipole of pKF1's centre in pKF2 (src/ORBmatcher.cc:1056-1063)
    template <class KeyFrameT>
    static void Epipole(KeyFrameT* pKF1, KeyFrameT* pKF2, float epf[2])
    {
        auto T2w = pKF2->GetPose();
        typedef Decay<decltype(T2w.translation())> Vec3;
        Vec3 Cw = pKF1->GetCameraCenter();
        Vec3 C2 = T2w * Cw;
        auto ep = pKF2->mpCamera->project(C2);
        epf[0] = ep(0); epf[1] = ep(1);
    }
    // ... and the fundamental matrix of Pinhole::epipolarConstrain (src/CameraModels/Pinhole.cpp:191-194), evaluated once instead of per pair
    template <class KeyFrameT>
    static void FundamentalAndEpipole(KeyFrameT* pKF1, KeyFrameT* pKF2, float f12[9], float epf[2])
    {
        Epipole(pKF1, pKF2, epf);
        auto T1w = pKF1->GetPose();
        auto Tw2 = pKF2->GetPoseInverse();
        typedef Decay<decltype(T1w.translation())> Vec3;
        typedef Decay<decltype(T1w.rotationMatrix())> Mat3;
        auto T12 = T1w * Tw2;
        Mat3 R12 = T12.rotationMatrix();
        Vec3 t12 = T12.translation();
        Mat3 t12x = R12;
        t12x(0,0) = 0; t12x(0,1) = -t12(2); t12x(0,2) = t12(1); t12x(1,0) = t12(2); t12x(1,1) = 0; t12x(1,2) = -t12(0); t12x(2,0) = -t12(1); t12x(2,1) = t12(0); t12x(2,2) = 0;
        Mat3 K1 = pKF1->mpCamera->toK_();
        Mat3 K2 = pKF2->mpCamera->toK_();
        Mat3 F12 = K1.transpose().inverse() * t12x * R12 * K2.inverse();
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) f12[r * 3 + c] = F12(r, c);
    }

    // Matching to triangulate new MapPoints. Check Epipolar Constraint.   — src/ORBmatcher.cc:1045-1323 (pinhole, one camera)
    template <class KeyFrameT>
    int SearchForTriangulation(KeyFrameT *pKF1, KeyFrameT* pKF2, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo, const bool bCoarse = false)
    {
        if (pKF1->N > 0 && !pKF1->mFeatVec.empty() && !pKF2->mFeatVec.empty()) {
            // both key frames stay on the device between calls (ImplicitCache): LocalMapping::CreateNewMapPoints asks for the same pKF1 against every
            // neighbour in a row, and a neighbour comes back for many key frames - nothing but the per-call flags and the pose-dependent matrices goes up
            std::vector<std::vector<std::pair<size_t, size_t> > > vv;
            const std::vector<int> counts = SearchForTriangulation(pKF1, std::vector<KeyFrameT*>(1, pKF2), ImplicitCache<KeyFrameT>(), vv, bOnlyStereo, bCoarse);
            vMatchedPairs.swap(vv[0]);
            return counts[0];
        }
        float f12[9], epf[2];
        if (pKF1->mpCamera->GetType() == 1 /* GeometricCamera::CAM_FISHEYE */) {
            Epipole(pKF1, pKF2, epf);
            return SearchForTriangulationFisheye(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse, epf[0], epf[1]);
        }
        if (pKF1->mpCamera2 || pKF2->mpCamera2) throw std::runtime_error("ORBmatcher (HIP): SearchForTriangulation on a rig of pinhole cameras is not supported (the reference only builds rigs of Kannala-Brandt cameras)");
        FundamentalAndEpipole(pKF1, pKF2, f12, epf);

        BowStore k1, k2;
        FillBow(*pKF1, pKF1->N, k1); FillBow(*pKF2, pKF2->N, k2);
        k1.present.assign(pKF1->N, 0); k2.present.assign(pKF2->N, 0);
        for (int i = 0; i < pKF1->N; i++) k1.present[i] = pKF1->GetMapPoint(i) != nullptr;
        for (int i = 0; i < pKF2->N; i++) k2.present[i] = pKF2->GetMapPoint(i) != nullptr;
        k1.v.has_map_point = k1.present.data(); k2.v.has_map_point = k2.present.data();
        std::vector<int> m12(pKF1->N > 0 ? pKF1->N : 1, -1);
        int nmatches = 0;
        {
            Check(orbm_search_for_triangulation(SharedHandle(), &k1.v, &k2.v, f12, epf, bOnlyStereo, bCoarse, mbCheckOrientation, m12.data(), &nmatches));
        }
        vMatchedPairs.clear();
        vMatchedPairs.reserve(nmatches);
        for (int i = 0; i < pKF1->N; i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
        return nmatches;
    }

    // The same for every neighbour of pKF1 in one call over device-resident key frames (LocalMapping::CreateNewMapPoints runs
    // SearchForTriangulation against 10-30 neighbours in a row, src/LocalMapping.cc:510-540; what a neighbour contributes - keys, descriptors,
    // mvuRight, mFeatVec - does not change while it lives and is uploaded once, into `cache`).  vvMatchedPairs[j] / the returned counts [j] are
    // what SearchForTriangulation(pKF1, vpNeighKFs[j], ...) gives (pinhole key frames, one Kannala-Brandt camera, or the Kannala-Brandt rig).
    template <class KeyFrameT>
    std::vector<int> SearchForTriangulation(KeyFrameT* pKF1, const std::vector<KeyFrameT*>& vpNeighKFs, ResidentKeyFrames<KeyFrameT>& cache,
                                            std::vector<std::vector<std::pair<size_t, size_t> > >& vvMatchedPairs, const bool bOnlyStereo, const bool bCoarse = false)
    {
        const int n2 = (int)vpNeighKFs.size(), N1 = pKF1->N;
        vvMatchedPairs.assign(n2, std::vector<std::pair<size_t, size_t> >());
        std::vector<int> counts(n2, 0);
        if (n2 == 0 || N1 == 0) return counts;
        cache.Trim();
        const bool fisheye = pKF1->mpCamera->GetType() == 1 /* GeometricCamera::CAM_FISHEYE */;
        std::vector<float> f12s((size_t)n2 * 9), eps((size_t)n2 * 2);
        std::vector<OrbmKB8Pair> kbs(fisheye ? n2 : 0);
        std::vector<orbm_keyframe*> k2(n2);
        std::vector<std::vector<uint8_t> > mp2(n2);
        std::vector<const uint8_t*> mp2p(n2);
        std::vector<uint8_t> mp1(N1);
        for (int i = 0; i < N1; i++) mp1[i] = pKF1->GetMapPoint(i) != nullptr;
        for (int j = 0; j < n2; j++) {
            KeyFrameT* pKF2 = vpNeighKFs[j];
            if (fisheye) { Epipole(pKF1, pKF2, &eps[(size_t)j * 2]); FillKB8Pair(pKF1, pKF2, kbs[j]); }
            else {
                if (pKF1->mpCamera2 || pKF2->mpCamera2) throw std::runtime_error("ORBmatcher (HIP): SearchForTriangulation on a rig of pinhole cameras is not supported (the reference only builds rigs of Kannala-Brandt cameras)");
                FundamentalAndEpipole(pKF1, pKF2, &f12s[(size_t)j * 9], &eps[(size_t)j * 2]);
            }
            k2[j] = cache.Get(pKF2);
            mp2[j].resize(pKF2->N > 0 ? pKF2->N : 1);
            for (int i = 0; i < pKF2->N; i++) mp2[j][i] = pKF2->GetMapPoint(i) != nullptr;
            mp2p[j] = mp2[j].data();
        }
        orbm_keyframe* k1 = cache.Get(pKF1);
        std::vector<int> m12((size_t)n2 * N1, -1);
        {
            if (fisheye) Check(orbm_search_for_triangulation_resident_kb8(SharedHandle(), k1, mp1.data(), n2, k2.data(), mp2p.data(), kbs.data(), eps.data(), bOnlyStereo, bCoarse,
                                                                          mbCheckOrientation, m12.data(), counts.data()));
            else Check(orbm_search_for_triangulation_resident(SharedHandle(), k1, mp1.data(), n2, k2.data(), mp2p.data(), f12s.data(), eps.data(), bOnlyStereo, bCoarse,
                                                              mbCheckOrientation, m12.data(), counts.data()));
        }
        for (int j = 0; j < n2; j++) {
            vvMatchedPairs[j].reserve(counts[j]);
            for (int i = 0; i < N1; i++) if (m12[(size_t)j * N1 + i] >= 0) vvMatchedPairs[j].push_back(std::make_pair((size_t)i, (size_t)m12[(size_t)j * N1 + i]));
        }
        return counts;
    }

    // cameras and relative poses of a pair of Kannala-Brandt key frames (src/ORBmatcher.cc:1067-1083)
    template <class KeyFrameT>
    static void FillKB8Pair(KeyFrameT* pKF1, KeyFrameT* pKF2, OrbmKB8Pair& kb)
    {
        const bool rig = pKF1->mpCamera2 && pKF2->mpCamera2;
        memset(&kb, 0, sizeof kb);
        kb.nleft1 = rig ? pKF1->NLeft : -1; kb.nleft2 = rig ? pKF2->NLeft : -1;
        for (int i = 0; i < 8; i++) {
            kb.cam1[0][i] = pKF1->mpCamera->getParameter(i); kb.cam2[0][i] = pKF2->mpCamera->getParameter(i);
            if (rig) { kb.cam1[1][i] = pKF1->mpCamera2->getParameter(i); kb.cam2[1][i] = pKF2->mpCamera2->getParameter(i); }
        }
        auto put = [&](int k, const decltype(pKF1->GetPose())& T) {
            const auto R = T.rotationMatrix(); const auto t = T.translation();
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) kb.R[k][3 * r + c] = R(r, c); kb.t[k][r] = t(r); }
        };
        auto T1w = pKF1->GetPose();
        auto Tw2 = pKF2->GetPoseInverse();
        if (!rig) put(0, T1w * Tw2);
        else {
            auto Tr1w = pKF1->GetRightPose();
            auto Twr2 = pKF2->GetRightPoseInverse();
            put(0, T1w * Tw2); put(1, T1w * Twr2); put(2, Tr1w * Tw2); put(3, Tr1w * Twr2);       // Tll, Tlr, Trl, Trr
        }
    }

    // Kannala-Brandt cameras (one fisheye camera, or the two-camera rig): the epipolar test is KannalaBrandt8::epipolarConstrain =
    // TriangulateMatches > 1e-4 with the relative pose of the pair of cameras the two features belong to (src/ORBmatcher.cc:1067-1083, :1203-1240)
    template <class KeyFrameT>
    int SearchForTriangulationFisheye(KeyFrameT *pKF1, KeyFrameT* pKF2, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo, const bool bCoarse, float epx, float epy)
    {
        OrbmKB8Pair kb; FillKB8Pair(pKF1, pKF2, kb);
        BowStore k1, k2;
        FillBow(*pKF1, pKF1->N, k1); FillBow(*pKF2, pKF2->N, k2);
        if (pKF1->NLeft != -1) FillAllKeys(*pKF1, pKF1->NLeft, pKF1->N, k1);
        if (pKF2->NLeft != -1) FillAllKeys(*pKF2, pKF2->NLeft, pKF2->N, k2);
        if (pKF1->mpCamera2) k1.v.u_right = nullptr;                    // bStereo1 = !mpCamera2 && mvuRight[idx1] >= 0 (:1126)
        if (pKF2->mpCamera2) k2.v.u_right = nullptr;
        k1.present.assign(pKF1->N > 0 ? pKF1->N : 1, 0); k2.present.assign(pKF2->N > 0 ? pKF2->N : 1, 0);
        for (int i = 0; i < pKF1->N; i++) k1.present[i] = pKF1->GetMapPoint(i) != nullptr;
        for (int i = 0; i < pKF2->N; i++) k2.present[i] = pKF2->GetMapPoint(i) != nullptr;
        k1.v.has_map_point = k1.present.data(); k2.v.has_map_point = k2.present.data();
        std::vector<int> m12(pKF1->N > 0 ? pKF1->N : 1, -1);
        const float epf[2] = {epx, epy};
        int nmatches = 0;
        {
            Check(orbm_search_for_triangulation_kb8(SharedHandle(), &k1.v, &k2.v, &kb, epf, bOnlyStereo, bCoarse, mbCheckOrientation, m12.data(), &nmatches));
        }
        vMatchedPairs.clear();
        vMatchedPairs.reserve(nmatches);
        for (int i = 0; i < pKF1->N; i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
        return nmatches;
    }

    // Search matches between MapPoints seen in KF1 and KF2 transforming by a Sim3 [s12*R12|t12]
    // In the stereo and RGB-D case, s12=1   — src/ORBmatcher.cc:1689-1932
    template <class KeyFrameT, class MapPointT, class Sim3T>
    int SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT *> &vpMatches12, const Sim3T &S12, const float th)
    {
        KeyFrameT* kf[2] = {pKF1, pKF2};
        const std::vector<MapPointT*> own[2] = {pKF1->GetMapPointMatches(), pKF2->GetMapPointMatches()};
        const int n[2] = {(int)own[0].size(), (int)own[1].size()};
        // points that already have a partner on either side take no part (:1718-1735)
        std::vector<uint8_t> paired[2] = {std::vector<uint8_t>(n[0] > 0 ? n[0] : 1, 0), std::vector<uint8_t>(n[1] > 0 ? n[1] : 1, 0)};
        for (int i = 0; i < n[0]; i++) {
            MapPointT* partner = vpMatches12[i];
            if (!partner) continue;
            paired[0][i] = 1;
            const int j = std::get<0>(partner->GetIndexInKeyFrame(pKF2));
            if (j >= 0 && j < n[1]) paired[1][j] = 1;
        }
        const Sim3T across[2] = {S12.inverse(), S12};          // KF1's points enter KF2 through S21, KF2's enter KF1 through S12
        ProjStore proj[2] = {ProjStore(n[0]), ProjStore(n[1])};
        for (int side = 0; side < 2; side++) {
            KeyFrameT* from = kf[side]; KeyFrameT* into = kf[1 - side];
            PointBatch pts(n[side]);
            for (int i = 0; i < n[side]; i++) {
                MapPointT* pMP = own[side][i];
                if (pMP && !paired[side][i] && !pMP->isBad()) pts.take(i, pMP, false);
            }
            // both directions use KF1's intrinsics, written out as x = X / Z, u = fx x + cx (:1760-1766, :1845-1851); the range is |p_c| (:1771)
            OrbmProjection spec = Spec(from->GetPose(), pKF1->mpCamera, *into, /*IsInImage*/ 1);
            spec.camera_type = 0; spec.cam[0] = pKF1->fx; spec.cam[1] = pKF1->fy; spec.cam[2] = pKF1->cx; spec.cam[3] = pKF1->cy; spec.inline_pinhole = 1;
            SpecSecond(spec, across[side]);
            spec.depth_test = 1; spec.dist_mode = 1; spec.distance_test = 1;
            pts.project(spec);
            for (int i = 0; i < n[side]; i++) {
                if (!pts.valid[i]) continue;
                proj[side].set(i, pts.u[i], pts.v[i], 0.0f, own[side][i]->PredictScale(pts.dist[i], into), 0.0f);
                CopyDescriptor(own[side][i], proj[side].descAt(i));
            }
        }
        FrameStore f1, f2; FillFrame(*pKF1, f1); FillFrame(*pKF2, f2);
        std::vector<int> m12(n[0] > 0 ? n[0] : 1, -1);
        int nFound = 0;
        {
            Check(orbm_search_by_sim3(SharedHandle(), &f1.v, &f2.v, proj[0].view(), proj[1].view(), th, m12.data(), &nFound));
        }
        for (int i = 0; i < n[0]; i++) if (m12[i] >= 0) vpMatches12[i] = own[1][m12[i]];
        return nFound;
    }

    // Project MapPoints into KeyFrame and search for duplicated MapPoints.   — src/ORBmatcher.cc:1325-1528
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT* pKF, const std::vector<MapPointT *> &vpMapPoints, const float th=3.0, const bool bRight = false)
    {
        const bool rig = pKF->NLeft != -1;
        if (bRight && !rig) throw std::runtime_error("ORBmatcher (HIP): Fuse(bRight) on a key frame without a second camera");
        const int nMPs = vpMapPoints.size();
        // geometry of every point that could reach the window search.  Whether it does is decided again in the replay loop below: the
        // reference's bad / IsInKeyFrame tests read state that earlier iterations of its loop may have changed (Replace, AddObservation).
        PointBatch pts(nMPs);
        for (int i = 0; i < nMPs; i++) {
            MapPointT* pMP = vpMapPoints[i];
            if (pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF)) pts.take(i, pMP, true);
        }
        OrbmProjection spec = bRight ? Spec(pKF->GetRightPose(), pKF->mpCamera2, *pKF, /*IsInImage*/ 1) : Spec(pKF->GetPose(), pKF->mpCamera, *pKF, 1);
        if (bRight) SpecCentre(spec, pKF->GetRightCameraCenter()); else SpecCentre(spec, pKF->GetCameraCenter());
        spec.depth_test = 1; spec.distance_test = 1; spec.angle_test = 1; spec.bf = pKF->mbf;
        pts.project(spec);
        ProjStore ps(nMPs);
        for (int i = 0; i < nMPs; i++) {
            if (!pts.valid[i]) continue;
            ps.set(i, pts.u[i], pts.v[i], pts.ur[i], vpMapPoints[i]->PredictScale(pts.dist[i], pKF), 0.0f);
            CopyDescriptor(vpMapPoints[i], ps.descAt(i));
        }
        FrameStore fs;
        if (!rig) FillFrame(*pKF, fs);
        else {                                              // one camera of the rig: its own keypoints, descriptor rows and grid (:1432-1434, :1468)
            const int n = bRight ? pKF->N - pKF->NLeft : pKF->NLeft;
            ConvertKeys(bRight ? pKF->mvKeysRight : pKF->mvKeys, n, fs.keys);
            fs.occ.assign(n > 0 ? n : 1, 0);
            fs.v.N = n; fs.v.keys_un = fs.keys.data(); fs.v.desc = pKF->mDescriptors.ptr(0) + (bRight ? (size_t)pKF->NLeft * 32 : 0); fs.v.u_right = nullptr;
            fs.v.occupied = fs.occ.data();
            FillBounds(*pKF, fs.v);
        }
        std::vector<int> best(nMPs > 0 ? nMPs : 1, -1);
        {
            Check(orbm_fuse_candidates(SharedHandle(), &fs.v, ps.view(), th, 1, pKF->mvInvLevelSigma2.data(), best.data(), nullptr));
        }
        if (bRight) for (int i = 0; i < nMPs; i++) if (best[i] >= 0) best[i] += pKF->NLeft;       // :1488
        // the map surgery of :1494-1520, one candidate after the other (an earlier merge can make a later point bad or put it into the key frame)
        int merged = 0;
        for (int i = 0; i < nMPs; i++) {
            MapPointT* incoming = vpMapPoints[i];
            if (!incoming || !ps.valid[i] || best[i] < 0 || incoming->isBad() || incoming->IsInKeyFrame(pKF)) continue;
            MapPointT* resident = Attach(pKF, incoming, best[i]);
            if (resident && !resident->isBad()) {           // two points for one keypoint: the one with more observations survives
                MapPointT* keep = resident->Observations() > incoming->Observations() ? resident : incoming;
                (keep == resident ? incoming : resident)->Replace(keep);
            }
            merged++;
        }
        return merged;
    }

    // Project MapPoints into KeyFrame using a given Sim3 and search for duplicated MapPoints.   — src/ORBmatcher.cc:1543-1660
    template <class KeyFrameT, class Sim3T, class MapPointT>
    int Fuse(KeyFrameT* pKF, Sim3T &Scw, const std::vector<MapPointT*> &vpPoints, float th, std::vector<MapPointT *> &vpReplacePoint)
    {
        const auto pose = RigidPart<Decay<decltype(pKF->GetPose())> >(Scw);
        const std::set<MapPointT*> inKeyFrame = pKF->GetMapPoints();
        const int nPoints = vpPoints.size();
        PointBatch pts(nPoints);
        for (int i = 0; i < nPoints; i++) {
            MapPointT* pMP = vpPoints[i];
            if (!pMP->isBad() && !inKeyFrame.count(pMP)) pts.take(i, pMP, true);
        }
        OrbmProjection spec = Spec(pose, pKF->mpCamera, *pKF, /*IsInImage*/ 1);
        SpecCentre(spec, pose.inverse().translation());
        spec.depth_test = 1; spec.distance_test = 1; spec.angle_test = 1;
        pts.project(spec);
        ProjStore ps(nPoints);
        for (int i = 0; i < nPoints; i++) {
            if (!pts.valid[i]) continue;
            ps.set(i, pts.u[i], pts.v[i], 0.0f, vpPoints[i]->PredictScale(pts.dist[i], pKF), 0.0f);
            CopyDescriptor(vpPoints[i], ps.descAt(i));
        }
        FrameStore fs; FillFrame(*pKF, fs);
        std::vector<int> best(nPoints > 0 ? nPoints : 1, -1);
        {
            Check(orbm_fuse_candidates(SharedHandle(), &fs.v, ps.view(), th, 0, nullptr, best.data(), nullptr));
        }
        int merged = 0;                                     // :1640-1656: here a keypoint that already has a point only reports it
        for (int i = 0; i < nPoints; i++) {
            if (!ps.valid[i] || best[i] < 0 || vpPoints[i]->isBad()) continue;
            MapPointT* resident = Attach(pKF, vpPoints[i], best[i]);
            if (resident && !resident->isBad()) vpReplacePoint[i] = resident;
            merged++;
        }
        return merged;
    }

public:
    // One library handle (HIP streams + device scratch) per calling THREAD: the reference's matchers are stateless and are called
    // concurrently from the Tracking, LocalMapping and LoopClosing threads; with a handle of its own each of them runs
    // its searches on its own stream without waiting for the others, and no lock is needed around a call (a handle is never shared).
    // The handle lives as long as its thread.
    // Multi-GPU hosts (one SLAM system per GPU, BASELINE.json configs[4]): the handle is created on the GPU chosen by SetDevice() - process-wide,
    // the default for every thread - or SetThreadDevice() - the calling thread only, e.g. the tracking thread of the system that owns GPU k;
    // pass the same index as the extractor's device_id.  Choose the device BEFORE a thread's first matcher call: changing it later re-creates the
    // thread's handle on its next call (SetDevice: every thread without a SetThreadDevice of its own) and empties its ResidentKeyFrames caches.
    static void SetDevice(int device) { ProcessDevice().store(device); }
    static void SetThreadDevice(int device) { ThreadDevice() = device; }
    static int Device() { const int t = ThreadDevice(); return t >= 0 ? t : ProcessDevice().load(); }
    static orbx_extractor* SharedHandle()
    {
        struct Holder { orbx_extractor* h = nullptr; int device = -1; ~Holder() { if (h) orbx_destroy(h); } };
        static thread_local Holder t;
        const int want = Device();
        if (t.h && t.device != want) { orbx_destroy(t.h); t.h = nullptr; }
        // (the extractor geometry of a matcher handle is never used: it only owns streams, staging buffers and the search kernels' scratch)
        if (!t.h) {
            if (orbx_create(&t.h, 1000, 1.2f, 8, 20, 7, want) != ORBX_OK) { t.h = nullptr; throw std::runtime_error(std::string("ORBmatcher (HIP): ") + orbx_last_error()); }
            t.device = want;
        }
        return t.h;
    }
private:
    static std::atomic<int>& ProcessDevice() { static std::atomic<int> d(0); return d; }
    static int& ThreadDevice() { static thread_local int d = -1; return d; }
public:
    static void Check(int rc) { if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher (HIP): ") + orbx_last_error()); }

protected:
    enum Occupancy { OccupiedAny, OccupiedWithObservations };

    struct FrameStore { std::vector<OrbxKeyPoint> keys; std::vector<uint8_t> occ; OrbmFrameView v; };
    struct RigStore { FrameStore l, r; OrbmFisheyeFrameView v; };
    struct BowStore {
        std::vector<OrbxKeyPoint> keys; std::vector<uint8_t> present; std::vector<uint32_t> node, feat; std::vector<int> start; OrbmKeyFrameView v;
    };
    // survivors of the host-side head of a projection loop
    struct ProjStore {
        std::vector<uint8_t> valid, desc; std::vector<float> u, v, ur, angle; std::vector<int> level; OrbmProjectedPointView pv;
        explicit ProjStore(int M) : valid(M > 0 ? M : 1, 0), desc((size_t)(M > 0 ? M : 1) * 32, 0), u(M > 0 ? M : 1, 0), v(M > 0 ? M : 1, 0), ur(M > 0 ? M : 1, 0),
                                    angle(M > 0 ? M : 1, 0), level(M > 0 ? M : 1, 0) { pv.M = M; }
        void set(int i, float uu, float vv, float r, int l, float a) { valid[i] = 1; u[i] = uu; v[i] = vv; ur[i] = r; level[i] = l; angle[i] = a; }
        uint8_t* descAt(int i) { return &desc[(size_t)i * 32]; }
        const OrbmProjectedPointView* view() {
            pv.valid = valid.data(); pv.u = u.data(); pv.v = v.data(); pv.ur = ur.data(); pv.pred_level = level.data(); pv.angle = angle.data(); pv.desc = desc.data();
            return &pv;
        }
    };

    // ---- the geometry in front of the window searches, on the device (orbm_project_points) ----
    // raw fields of the map points of one call, read through the caller's accessors; entries never taken stay skipped
    struct PointBatch {
        int M;
        std::vector<float> pos, normal, minInv, maxInv, u, v, ur, invz, dist; std::vector<uint8_t> skip, valid;
        explicit PointBatch(int m) : M(m), pos(3 * (size_t)(m > 0 ? m : 1), 0.f), normal(3 * (size_t)(m > 0 ? m : 1), 0.f), minInv(m > 0 ? m : 1, 0.f), maxInv(m > 0 ? m : 1, 0.f),
                                     u(m > 0 ? m : 1, 0.f), v(m > 0 ? m : 1, 0.f), ur(m > 0 ? m : 1, 0.f), invz(m > 0 ? m : 1, 0.f), dist(m > 0 ? m : 1, 0.f), skip(m > 0 ? m : 1, 1),
                                     valid(m > 0 ? m : 1, 0) {}
        template <class MapPointT> void take(int i, MapPointT* p, bool withNormal)
        {
            const auto w = p->GetWorldPos();
            for (int k = 0; k < 3; k++) pos[3 * (size_t)i + k] = w(k);
            if (withNormal) { const auto nrm = p->GetNormal(); for (int k = 0; k < 3; k++) normal[3 * (size_t)i + k] = nrm(k); }
            minInv[i] = p->GetMinDistanceInvariance(); maxInv[i] = p->GetMaxDistanceInvariance();
            skip[i] = 0;
        }
        void project(const OrbmProjection& spec)
        {
            OrbmProjectIn in = {M, pos.data(), normal.data(), minInv.data(), maxInv.data(), skip.data()};
            OrbmProjectOut out = {valid.data(), u.data(), v.data(), ur.data(), invz.data(), dist.data()};
            Check(orbm_project_points(SharedHandle(), &spec, &in, &out));
        }
    };
    // pose, camera and image test of a projection; the tests a method applies are switched on by the method.  The pose enters as Sophus holds
    // it - unit quaternion + translation - because `Tcw * p3Dw` is Sophus' quaternion action (so3.hpp:357-367), not a matrix product
    template <class PoseT, class CameraT, class HolderT> static OrbmProjection Spec(const PoseT& T, CameraT* camera, HolderT& image, int boundsMode)
    {
        OrbmProjection s; memset(&s, 0, sizeof s);
        const auto& q = T.unit_quaternion(); const auto t = T.translation();
        s.q[0] = q.x(); s.q[1] = q.y(); s.q[2] = q.z(); s.q[3] = q.w();
        for (int r = 0; r < 3; r++) s.t[r] = t(r);
        s.camera_type = camera->GetType() == 1 ? 1 : 0;
        for (int i = 0; i < (s.camera_type ? 8 : 4); i++) s.cam[i] = camera->getParameter(i);
        s.min_x = image.mnMinX; s.max_x = image.mnMaxX; s.min_y = image.mnMinY; s.max_y = image.mnMaxY; s.bounds_mode = boundsMode;
        return s;
    }
    template <class Vec3T> static void SpecCentre(OrbmProjection& s, const Vec3T& c) { for (int k = 0; k < 3; k++) s.Ow[k] = c(k); }
    // a second transform behind the pose: a Sim3 (its quaternion carries the scale: |q|^2 = scale(); rxso3.hpp:265-273) or an SE3
    template <class T> static auto SecondQuaternion(const T& x, OrbmProjection& s, int) -> decltype(x.scale(), void())
    {
        const auto& q = x.quaternion(); s.q2[0] = q.x(); s.q2[1] = q.y(); s.q2[2] = q.z(); s.q2[3] = q.w();
        s.s2 = x.scale(); s.second = 1;              // scale() IS quaternion().squaredNorm() (rxso3.hpp:350), evaluated by the caller's own Eigen
    }
    template <class T> static void SecondQuaternion(const T& x, OrbmProjection& s, long)
    {
        const auto& q = x.unit_quaternion(); s.q2[0] = q.x(); s.q2[1] = q.y(); s.q2[2] = q.z(); s.q2[3] = q.w();
        s.s2 = 1.0f; s.second = 2;
    }
    template <class TransformT> static void SpecSecond(OrbmProjection& s, const TransformT& X)
    {
        SecondQuaternion(X, s, 0);
        const auto t = X.translation();
        for (int r = 0; r < 3; r++) s.t2[r] = t(r);
    }
    // the rigid transform a similarity moves points with up to scale: [R | t / s] (:508, :1559)
    template <class SE3T, class Sim3T> static SE3T RigidPart(const Sim3T& S) { return SE3T(S.rotationMatrix(), S.translation() / S.scale()); }

    // gives keypoint `slot` of the key frame the map point unless it has one already; returns the point that is there (NULL: `point` went in)
    template <class KeyFrameT, class MapPointT> static MapPointT* Attach(KeyFrameT* kf, MapPointT* point, int slot)
    {
        MapPointT* there = kf->GetMapPoint(slot);
        if (!there) { point->AddObservation(kf, slot); kf->AddMapPoint(point, slot); }
        return there;
    }

    template <class MapPointT> static void CopyDescriptor(MapPointT* p, uint8_t* dst) { const cv::Mat d = p->GetDescriptor(); memcpy(dst, d.ptr(0), 32); }

    static void ConvertKeys(const std::vector<cv::KeyPoint>& in, int n, std::vector<OrbxKeyPoint>& out)
    {
        out.resize(n > 0 ? n : 1);
        for (int i = 0; i < n; i++) {
            const cv::KeyPoint& k = in[i];
            out[i].x = k.pt.x; out[i].y = k.pt.y; out[i].size = k.size; out[i].angle = k.angle; out[i].response = k.response; out[i].octave = k.octave; out[i].class_id = k.class_id;
        }
    }
    template <class H> static void FillBounds(H& F, OrbmFrameView& fv)
    {
        fv.min_x = F.mnMinX; fv.min_y = F.mnMinY; fv.max_x = F.mnMaxX; fv.max_y = F.mnMaxY;
        fv.grid_w_inv = F.mfGridElementWidthInv; fv.grid_h_inv = F.mfGridElementHeightInv; fv.mbf = F.mbf;
        fv.nlevels = (int)F.mvScaleFactors.size(); fv.scale_factors = F.mvScaleFactors.data();
    }
    // Frame::mvpMapPoints is public, KeyFrame's is not: occupancy is only ever read from Frames
    template <class FrameT> static void FillOccupancy(FrameT& F, int first, int n, Occupancy occ, std::vector<uint8_t>& out)
    {
        out.assign(n > 0 ? n : 1, 0);
        for (int i = 0; i < n; i++) if (F.mvpMapPoints[first + i]) out[i] = occ == OccupiedAny ? 1 : F.mvpMapPoints[first + i]->Observations() > 0;
    }
    // one-camera Frame or KeyFrame as the target of a window search (nothing occupied)
    template <class H> static void FillFrame(H& F, FrameStore& s)
    {
        const int N = (int)F.mvKeysUn.size();
        ConvertKeys(F.mvKeysUn, N, s.keys);
        s.occ.assign(N > 0 ? N : 1, 0);
        s.v.N = N; s.v.keys_un = s.keys.data(); s.v.desc = F.mDescriptors.ptr(0); s.v.u_right = F.mvuRight.empty() ? nullptr : F.mvuRight.data();
        s.v.occupied = s.occ.data();
        FillBounds(F, s.v);
    }
    template <class FrameT> static void FillFrame(FrameT& F, FrameStore& s, Occupancy occ)
    {
        FillFrame(F, s);
        FillOccupancy(F, 0, s.v.N, occ, s.occ);
        s.v.occupied = s.occ.data();
    }
    // fisheye-rig Frame: camera 1 = mvKeys / rows [0, Nleft), camera 2 = mvKeysRight / rows [Nleft, N)
    template <class FrameT> static void FillRig(FrameT& F, RigStore& s, Occupancy occ)
    {
        const int NL = F.Nleft, NR = F.N - F.Nleft;
        ConvertKeys(F.mvKeys, NL, s.l.keys); ConvertKeys(F.mvKeysRight, NR, s.r.keys);
        FillOccupancy(F, 0, NL, occ, s.l.occ); FillOccupancy(F, NL, NR, occ, s.r.occ);
        s.l.v.N = NL; s.l.v.keys_un = s.l.keys.data(); s.l.v.desc = F.mDescriptors.ptr(0); s.l.v.u_right = nullptr; s.l.v.occupied = s.l.occ.data();
        s.r.v.N = NR; s.r.v.keys_un = s.r.keys.data(); s.r.v.desc = F.mDescriptors.ptr(0) + (size_t)NL * 32; s.r.v.u_right = nullptr; s.r.v.occupied = s.r.occ.data();
        FillBounds(F, s.l.v); FillBounds(F, s.r.v);
        s.v.left = s.l.v; s.v.right = s.r.v;
        s.v.left_to_right = F.mvLeftToRightMatch.empty() ? nullptr : F.mvLeftToRightMatch.data();
        s.v.right_to_left = F.mvRightToLeftMatch.empty() ? nullptr : F.mvRightToLeftMatch.data();
    }
    // Frame or KeyFrame with its DBoW2::FeatureVector (map<NodeId, vector<unsigned>>) flattened to CSR
    // keypoints of a rig Frame / KeyFrame by feature index (mvKeys, then mvKeysRight); a one-camera holder keeps mvKeysUn
    template <class H> static void FillAllKeys(H& K, int nleft, int N, BowStore& s)
    {
        if (nleft == -1) return;
        std::vector<OrbxKeyPoint> r;
        ConvertKeys(K.mvKeys, nleft, s.keys); s.keys.resize(nleft);
        ConvertKeys(K.mvKeysRight, N - nleft, r); r.resize(N - nleft);
        s.keys.insert(s.keys.end(), r.begin(), r.end());
        if (s.keys.empty()) s.keys.resize(1);
        s.v.keys_un = s.keys.data();
    }
    template <class H> static void FillBow(H& K, int N, BowStore& s)
    {
        ConvertKeys(K.mvKeysUn, (int)K.mvKeysUn.size() < N ? (int)K.mvKeysUn.size() : N, s.keys);
        if ((int)s.keys.size() < N) s.keys.resize(N);
        s.node.clear(); s.feat.clear(); s.start.assign(1, 0);
        for (auto it = K.mFeatVec.begin(); it != K.mFeatVec.end(); ++it) {
            s.node.push_back((uint32_t)it->first);
            for (size_t j = 0; j < it->second.size(); j++) s.feat.push_back((uint32_t)it->second[j]);
            s.start.push_back((int)s.feat.size());
        }
        if (s.node.empty()) s.node.push_back(0);
        if (s.feat.empty()) s.feat.push_back(0);
        s.v.N = N; s.v.keys_un = s.keys.data(); s.v.desc = K.mDescriptors.ptr(0); s.v.u_right = K.mvuRight.empty() ? nullptr : K.mvuRight.data();
        s.v.has_map_point = nullptr;
        s.v.fv_nodes = (int)s.start.size() - 1; s.v.fv_node_id = s.node.data(); s.v.fv_start = s.start.data(); s.v.fv_feat = s.feat.data();
        s.v.nlevels = (int)K.mvScaleFactors.size(); s.v.scale_factors = K.mvScaleFactors.data(); s.v.level_sigma2 = K.mvLevelSigma2.data();
    }

    // common body of the two SearchByProjection(KeyFrame*, Sim3, ...) overloads; they differ in how the projection is written
    // (:534 GeometricCamera::project, :656-660 inline pinhole arithmetic with pKF->fx ...)
    template <class KeyFrameT, class Sim3T, class MapPointT>
    int SearchBySim3Projection(KeyFrameT* pKF, Sim3T &Scw, const std::vector<MapPointT*> &vpPoints, std::vector<MapPointT*> &vpMatched, int th, float ratioHamming,
                               bool inlineProjection, std::vector<int>& assigned)
    {
        const auto pose = RigidPart<Decay<decltype(pKF->GetPose())> >(Scw);
        std::set<MapPointT*> taken(vpMatched.begin(), vpMatched.end());
        taken.erase(static_cast<MapPointT*>(NULL));
        const int M = (int)vpPoints.size();
        PointBatch pts(M);
        for (int i = 0; i < M; i++) {
            MapPointT* pMP = vpPoints[i];
            if (!pMP->isBad() && !taken.count(pMP)) pts.take(i, pMP, true);
        }
        OrbmProjection spec = Spec(pose, pKF->mpCamera, *pKF, /*IsInImage*/ 1);
        if (inlineProjection) { spec.camera_type = 0; spec.cam[0] = pKF->fx; spec.cam[1] = pKF->fy; spec.cam[2] = pKF->cx; spec.cam[3] = pKF->cy; spec.inline_pinhole = 1; }
        SpecCentre(spec, pose.inverse().translation());
        spec.depth_test = 1; spec.distance_test = 1; spec.angle_test = 1;
        pts.project(spec);
        ProjStore ps(M);
        for (int i = 0; i < M; i++) {
            if (!pts.valid[i]) continue;
            ps.set(i, pts.u[i], pts.v[i], 0.0f, vpPoints[i]->PredictScale(pts.dist[i], pKF), 0.0f);
            CopyDescriptor(vpPoints[i], ps.descAt(i));
        }
        FrameStore fs; FillFrame(*pKF, fs);
        const int N = (int)vpMatched.size();
        fs.occ.assign(N > 0 ? N : 1, 0);
        for (int i = 0; i < N; i++) fs.occ[i] = vpMatched[i] != nullptr;
        fs.v.occupied = fs.occ.data();
        assigned.assign(N > 0 ? N : 1, -1);
        int nmatches = 0;
        Check(orbm_search_by_projection_sim3(SharedHandle(), &fs.v, ps.view(), (float)th, ratioHamming, assigned.data(), &nmatches));
        return nmatches;
    }

    float mfNNratio;
    bool mbCheckOrientation;
};

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

// Frame::ComputeStereoFromRGBD (src/Frame.cc:1361-1391) for a Frame whose extractor is the HIP facade: mvDepth / mvuRight of the keypoints of the
// LAST call of the extractor from the depth image (CV_32F, already scaled by the depth map factor); mvKeysUn is taken to equal mvKeys (no
// distortion), as the device never sees UndistortKeyPoints.  Call it where the reference calls the member:
//   Frame.cc:270   ComputeStereoFromRGBD(imDepth);   ->   ORB_SLAM3::ComputeStereoFromRGBD(*this, mpORBextractorLeft, imDepth);
template <class FrameT, class ExtractorT>
void ComputeStereoFromRGBD(FrameT& F, ExtractorT* pExtractor, const cv::Mat& imDepth)
{
    const int cap = orbx_max_keypoints(pExtractor->Handle());
    std::vector<float> u(cap, -1.0f), d(cap, -1.0f);
    int n = 0;
    ORBmatcher::Check(orbm_stereo_from_depth(pExtractor->Handle(), 0, 1, imDepth.template ptr<float>(0), (int)(imDepth.step / sizeof(float)),
                                             (size_t)(imDepth.step / sizeof(float)) * imDepth.rows, 0, F.mbf));
    ORBmatcher::Check(orbm_stereo_fetch(pExtractor->Handle(), 1, u.data(), d.data(), cap, &n));
    F.mvuRight.assign(u.begin(), u.begin() + F.N);
    F.mvDepth.assign(d.begin(), d.begin() + F.N);
}

// Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1530-1587) for a fisheye-rig Frame whose two extractors are the HIP facades: 2-NN + ratio
// test on the lapping keypoints and the triangulation gate (KannalaBrandt8::TriangulateMatches) on the device-resident results of the LAST
// call of each extractor (both made with the cameras' lapping areas, as ExtractORB does).  Fills mvLeftToRightMatch, mvRightToLeftMatch,
// mvDepth, mvuRight, mvStereo3Dpoints and resets mnCloseMPs, like the member function.  Call it where the reference calls the member:
//   Frame.cc:1511   ComputeStereoFishEyeMatches();   ->   ORB_SLAM3::ComputeStereoFishEyeMatches(*this, mpORBextractorLeft, mpORBextractorRight);
template <class FrameT, class ExtractorT>
void ComputeStereoFishEyeMatches(FrameT& F, ExtractorT* pLeft, ExtractorT* pRight)
{
    OrbmKB8Stereo cams;
    for (int i = 0; i < 8; i++) { cams.cam1[i] = F.mpCamera->getParameter(i); cams.cam2[i] = F.mpCamera2->getParameter(i); }
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) cams.R12[3 * r + c] = F.mRlr(r, c); cams.t12[r] = F.mtlr(r); }
    const int cap = orbx_max_keypoints(pLeft->Handle());
    std::vector<int> l2r(cap, -1), r2l(cap, -1);
    std::vector<float> depth(cap, -1.0f), p3d((size_t)cap * 3, 0.0f);
    int n = 0;
    ORBmatcher::Check(orbm_stereo_fisheye(pLeft->Handle(), 0, pRight->Handle(), 0, 1, &cams));
    ORBmatcher::Check(orbm_stereo_fisheye_fetch(pLeft->Handle(), 1, l2r.data(), r2l.data(), depth.data(), p3d.data(), &n, cap));
    F.mvLeftToRightMatch.assign(l2r.begin(), l2r.begin() + F.Nleft);
    F.mvRightToLeftMatch.assign(r2l.begin(), r2l.begin() + F.Nright);
    F.mvDepth.assign(depth.begin(), depth.begin() + F.Nleft);
    F.mvuRight = std::vector<float>(F.Nleft, -1);
    typedef typename std::decay<decltype(F.mvStereo3Dpoints[0])>::type Vec3;
    F.mvStereo3Dpoints = std::vector<Vec3>(F.Nleft);
    for (int i = 0; i < F.Nleft; i++) if (l2r[i] >= 0) F.mvStereo3Dpoints[i] = Vec3(p3d[3 * (size_t)i], p3d[3 * (size_t)i + 1], p3d[3 * (size_t)i + 2]);
    F.mnCloseMPs = 0;
}

} // namespace ORB_SLAM

#endif // ORB_SLAM3_AMD_ORBMATCHER_H
