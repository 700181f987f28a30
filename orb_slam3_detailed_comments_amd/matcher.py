"""Host-side mirror of the reference's descriptor matchers on top of the HIP library:
``ORBmatcher::DescriptorDistance`` (src/ORBmatcher.cc:2383-2403), ``Frame::ComputeStereoMatches``
(src/Frame.cc:1102-1358) and the kNN + ratio part of ``Frame::ComputeStereoFishEyeMatches`` (:1553-1562)."""
import ctypes as C

import numpy as np

from .sophus import SE3f, Sim3f, as_se3


class ORBmatcher:
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30   # src/ORBmatcher.cc:35-37

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)

    def SearchByProjection(self, ext, frame, map_points, th=1.0, bFarPoints=False, thFarPoints=50.0):
        """ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints), src/ORBmatcher.cc:45.
        frame / map_points: views.frame_view(...) / views.map_point_view(...).  Returns (nmatches, assigned[N]) where
        assigned[i] is the index of the map point written to F.mvpMapPoints[i] (-1: untouched)."""
        N = frame.view.N
        assigned = np.full(N, -1, np.int32); nm = C.c_int()
        ext._lib.check(ext._lib.L.orbm_search_by_projection_mappoints(ext._h, frame.ref(), map_points.ref(), float(th), int(bFarPoints),
                                                                    float(thFarPoints), self.mfNNratio, assigned.ctypes.data, C.byref(nm)))
        return nm.value, assigned

    def SearchByProjectionFrame(self, ext, cur, last, th, bForward=False, bBackward=False):
        """ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono), src/ORBmatcher.cc:1950.
        Returns (nmatches, assigned[N]): index into the last frame, -1 untouched, -2 reset by the rotation check."""
        N = cur.view.N
        assigned = np.full(N, -1, np.int32); nm = C.c_int()
        ext._lib.check(ext._lib.L.orbm_search_by_projection_frame(ext._h, cur.ref(), last.ref(), float(th), int(bForward), int(bBackward),
                                                                int(self.mbCheckOrientation), assigned.ctypes.data, C.byref(nm)))
        return nm.value, assigned

    def SearchForTriangulation(self, ext, kf1, kf2, F12, ep, bOnlyStereo=False, bCoarse=False):
        """ORBmatcher::SearchForTriangulation, src/ORBmatcher.cc:1045 (pinhole).  Returns (nmatches, vMatchedPairs [(i1,i2)...])."""
        F12 = np.ascontiguousarray(F12, np.float32).reshape(9); ep = np.ascontiguousarray(ep, np.float32).reshape(2)
        m12 = np.full(kf1.view.N, -1, np.int32); nm = C.c_int()
        ext._lib.check(ext._lib.L.orbm_search_for_triangulation(ext._h, kf1.ref(), kf2.ref(), F12.ctypes.data, ep.ctypes.data, int(bOnlyStereo),
                                                              int(bCoarse), int(self.mbCheckOrientation), m12.ctypes.data, C.byref(nm)))
        idx = np.nonzero(m12 >= 0)[0]
        return nm.value, [(int(i), int(m12[i])) for i in idx]

    def SearchForTriangulationBatch(self, ext, kf1, kf2s, F12s, eps, bOnlyStereo=False, bCoarse=False):
        """SearchForTriangulation of kf1 against every key frame of kf2s in one launch.  Returns a list of (nmatches, pairs) like
        SearchForTriangulation."""
        n2 = len(kf2s)
        F = np.ascontiguousarray(F12s, np.float32).reshape(n2, 9); E = np.ascontiguousarray(eps, np.float32).reshape(n2, 2)
        ptrs = (C.c_void_p * max(n2, 1))(*[C.cast(k.ref(), C.c_void_p) for k in kf2s])
        N1 = kf1.view.N
        m12 = np.full((max(n2, 1), max(N1, 1)), -1, np.int32); nm = np.zeros(max(n2, 1), np.int32)
        ext._lib.check(ext._lib.L.orbm_search_for_triangulation_batch(ext._h, kf1.ref(), n2, ptrs, F.ctypes.data, E.ctypes.data, int(bOnlyStereo), int(bCoarse),
                                                                    int(self.mbCheckOrientation), m12.ctypes.data, nm.ctypes.data))
        return [(int(nm[j]), [(int(i), int(m12[j, i])) for i in np.nonzero(m12[j, :N1] >= 0)[0]]) for j in range(n2)]

    def SearchByBoW(self, ext, kf1, kf2, frame_version=True):
        """ORBmatcher::SearchByBoW: frame_version=True is (KeyFrame*, Frame&, vpMapPointMatches), src/ORBmatcher.cc:259;
        False is (KeyFrame*, KeyFrame*, vpMatches12), :892.  Returns (nmatches, matches12[N1])."""
        m12 = np.full(kf1.view.N, -1, np.int32); nm = C.c_int()
        ext._lib.check(ext._lib.L.orbm_search_by_bow(ext._h, kf1.ref(), kf2.ref(), self.mfNNratio, int(frame_version), int(self.mbCheckOrientation),
                                                   m12.ctypes.data, C.byref(nm)))
        return nm.value, m12

    def SearchByBoWBatch(self, ext, kf1s, kf2s, frame_version=True):
        """SearchByBoW for the pairs (kf1s[p], kf2s[p]) in one launch (relocalisation candidates vs the current frame, a key frame vs its loop
        candidates).  Returns a list of (nmatches, matches12) like SearchByBoW."""
        n = len(kf1s)
        p1 = (C.c_void_p * max(n, 1))(*[C.cast(k.ref(), C.c_void_p) for k in kf1s]); p2 = (C.c_void_p * max(n, 1))(*[C.cast(k.ref(), C.c_void_p) for k in kf2s])
        outs = [np.full(max(k.view.N, 1), -1, np.int32) for k in kf1s]
        po = (C.c_void_p * max(n, 1))(*[o.ctypes.data for o in outs]); nm = np.zeros(max(n, 1), np.int32)
        ext._lib.check(ext._lib.L.orbm_search_by_bow_batch(ext._h, n, p1, p2, self.mfNNratio, int(frame_version), int(self.mbCheckOrientation), po, nm.ctypes.data))
        return [(int(nm[p]), outs[p][:kf1s[p].view.N]) for p in range(n)]

    def SearchForTriangulationResident(self, ext, kf1, mp1, kf2s, mp2s, F12s, eps, bOnlyStereo=False, bCoarse=False):
        """SearchForTriangulationBatch over ResidentKeyFrame objects: mp1 / mp2s[j] = uint8 flags "feature has a map point" at call time (None =
        none).  Returns the same list of (nmatches, pairs)."""
        n2 = len(kf2s)
        F = np.ascontiguousarray(F12s, np.float32).reshape(n2, 9); E = np.ascontiguousarray(eps, np.float32).reshape(n2, 2)
        ptrs = (C.c_void_p * max(n2, 1))(*[k._kf for k in kf2s])
        m1 = None if mp1 is None else np.ascontiguousarray(mp1, np.uint8)
        m2 = [None if m is None else np.ascontiguousarray(m, np.uint8) for m in mp2s]
        pm2 = (C.c_void_p * max(n2, 1))(*[None if m is None else m.ctypes.data for m in m2])
        N1 = kf1.N
        m12 = np.full((max(n2, 1), max(N1, 1)), -1, np.int32); nm = np.zeros(max(n2, 1), np.int32)
        ext._lib.check(ext._lib.L.orbm_search_for_triangulation_resident(ext._h, kf1._kf, None if m1 is None else m1.ctypes.data, n2, ptrs, pm2, F.ctypes.data,
                                                                       E.ctypes.data, int(bOnlyStereo), int(bCoarse), int(self.mbCheckOrientation), m12.ctypes.data,
                                                                       nm.ctypes.data))
        m12 = m12.reshape(-1)[:n2 * N1].reshape(n2, N1) if N1 > 0 else m12[:n2, :0]
        return [(int(nm[j]), [(int(i), int(m12[j, i])) for i in np.nonzero(m12[j] >= 0)[0]]) for j in range(n2)]

    def SearchByBoWResident(self, ext, kf1s, mp1s, kf2s, elig2s, frame_version=True):
        """SearchByBoWBatch over ResidentKeyFrame objects: mp1s[p] = uint8 flags "feature of kf1s[p] has a good map point", elig2s[p] = flags of the
        features of kf2s[p] that may be matched (None = all).  Returns the same list of (nmatches, matches12)."""
        n = len(kf1s)
        p1 = (C.c_void_p * max(n, 1))(*[k._kf for k in kf1s]); p2 = (C.c_void_p * max(n, 1))(*[k._kf for k in kf2s])
        a1 = [None if m is None else np.ascontiguousarray(m, np.uint8) for m in mp1s]; a2 = [None if m is None else np.ascontiguousarray(m, np.uint8) for m in elig2s]
        q1 = (C.c_void_p * max(n, 1))(*[None if m is None else m.ctypes.data for m in a1]); q2 = (C.c_void_p * max(n, 1))(*[None if m is None else m.ctypes.data for m in a2])
        outs = [np.full(max(k.N, 1), -1, np.int32) for k in kf1s]
        po = (C.c_void_p * max(n, 1))(*[o.ctypes.data for o in outs]); nm = np.zeros(max(n, 1), np.int32)
        ext._lib.check(ext._lib.L.orbm_search_by_bow_resident(ext._h, n, p1, q1, p2, q2, self.mfNNratio, int(frame_version), int(self.mbCheckOrientation), po, nm.ctypes.data))
        return [(int(nm[p]), outs[p][:kf1s[p].N]) for p in range(n)]

    def SearchByBoWFramesBatch(self, ext, voc, kfs, mps, first=0):
        """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) for the frames [first, first + len(kfs)) of ext's last extraction on the device
        (orbm_search_by_bow_frames_batch): voc = the ORBVocabulary whose transform_extracted(ext, first, B, levelsup) has run on these frames, kfs[b] = the
        ResidentKeyFrame frame b is searched against, mps[b] = uint8 flags "feature of kfs[b] has a good map point".  Returns [(nmatches, matches12)]
        with matches12[i] = frame feature matched to key-frame feature i or -1."""
        B = len(kfs)
        p1 = (C.c_void_p * max(B, 1))(*[k._kf for k in kfs])
        a1 = [np.ascontiguousarray(m, np.uint8) for m in mps]
        q1 = (C.c_void_p * max(B, 1))(*[m.ctypes.data for m in a1])
        outs = [np.full(max(k.N, 1), -1, np.int32) for k in kfs]
        po = (C.c_void_p * max(B, 1))(*[o.ctypes.data for o in outs]); nm = np.zeros(max(B, 1), np.int32)
        ext._lib.check(ext._lib.L.orbm_search_by_bow_frames_batch(ext._h, voc._v, int(first), B, p1, q1, self.mfNNratio, int(self.mbCheckOrientation), po, nm.ctypes.data))
        return [(int(nm[b]), outs[b][:kfs[b].N]) for b in range(B)]

    def SearchByBoWFisheye(self, ext, kf, frame, nleft):
        """SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) for a fisheye-rig frame (F.Nleft = nleft != -1), src/ORBmatcher.cc:259-493.
        Both views list all features by index (camera 1 first).  Returns (nmatches, assigned[N_frame] = key-frame feature or -1)."""
        a2 = np.full(max(frame.view.N, 1), -1, np.int32); nm = C.c_int()
        ext._lib.check(ext._lib.L.orbm_search_by_bow_fisheye(ext._h, kf.ref(), frame.ref(), int(nleft), self.mfNNratio, int(self.mbCheckOrientation),
                                                           a2.ctypes.data, C.byref(nm)))
        return nm.value, a2[:frame.view.N]

    def SearchForInitialization(self, ext, f1, f2, vbPrevMatched, windowSize=10):
        """ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:734.  vbPrevMatched [N1,2] float32 is updated in place.
        Returns (nmatches, vnMatches12)."""
        assert vbPrevMatched.dtype == np.float32 and vbPrevMatched.flags["C_CONTIGUOUS"]
        m12 = np.full(f1.view.N, -1, np.int32); nm = C.c_int()
        ext._lib.check(ext._lib.L.orbm_search_for_initialization(ext._h, f1.ref(), f2.ref(), vbPrevMatched.ctypes.data, int(windowSize), self.mfNNratio,
                                                               int(self.mbCheckOrientation), m12.ctypes.data, C.byref(nm)))
        return nm.value, m12

    def SearchByProjectionFisheye(self, ext, frame2, map_points, map_points_r, th=1.0, bFarPoints=False, thFarPoints=50.0):
        """SearchByProjection(Frame&, vector<MapPoint*>&, ...) for a two-camera frame (Nleft != -1), src/ORBmatcher.cc:45-239.
        frame2: views.fisheye_frame_view; map_points_r: views.map_point_right_view.  Returns (nmatches, assigned[Nleft + Nright])."""
        N = frame2.view.left.N + frame2.view.right.N
        assigned = np.full(N, -1, np.int32); nm = C.c_int()
        ext._lib.check(ext._lib.L.orbm_search_by_projection_mappoints_fisheye(ext._h, frame2.ref(), map_points.ref(), map_points_r.ref(), float(th),
                                                                            int(bFarPoints), float(thFarPoints), self.mfNNratio, assigned.ctypes.data, C.byref(nm)))
        return nm.value, assigned

    def SearchByProjectionFrameFisheye(self, ext, cur2, last, proj_ur, proj_vr, th, bForward=False, bBackward=False):
        """SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) for a two-camera current frame, src/ORBmatcher.cc:1950-2184."""
        N = cur2.view.left.N + cur2.view.right.N
        ur = np.ascontiguousarray(proj_ur, np.float32); vr = np.ascontiguousarray(proj_vr, np.float32)
        assigned = np.full(N, -1, np.int32); nm = C.c_int()
        ext._lib.check(ext._lib.L.orbm_search_by_projection_frame_fisheye(ext._h, cur2.ref(), last.ref(), ur.ctypes.data, vr.ctypes.data, float(th),
                                                                        int(bForward), int(bBackward), int(self.mbCheckOrientation), assigned.ctypes.data, C.byref(nm)))
        return nm.value, assigned

    def SearchByProjectionSim3(self, ext, kf, points, th, ratioHamming=1.0):
        """ORBmatcher::SearchByProjection(KeyFrame*, Sim3f&, vpPoints, [vpPointsKFs,] vpMatched, [vpMatchedKF,] th, ratioHamming),
        src/ORBmatcher.cc:495 and :608.  kf: views.frame_view with occupied = (vpMatched[idx] != NULL); points:
        views.projected_point_view.  Returns (nmatches, assigned[N]) with assigned[idx] = index of the point put in vpMatched[idx]."""
        assigned = np.full(kf.view.N, -1, np.int32); nm = C.c_int()
        ext._lib.check(ext._lib.L.orbm_search_by_projection_sim3(ext._h, kf.ref(), points.ref(), float(th), float(ratioHamming),
                                                               assigned.ctypes.data, C.byref(nm)))
        return nm.value, assigned

    def SearchByProjectionKeyFrame(self, ext, cur, points, th, ORBdist):
        """ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:2196.
        cur.occupied = (CurrentFrame.mvpMapPoints[i] != NULL).  Returns (nmatches, assigned[N]) (-2: reset by the rotation check)."""
        assigned = np.full(cur.view.N, -1, np.int32); nm = C.c_int()
        ext._lib.check(ext._lib.L.orbm_search_by_projection_keyframe(ext._h, cur.ref(), points.ref(), float(th), int(ORBdist),
                                                                   int(self.mbCheckOrientation), assigned.ctypes.data, C.byref(nm)))
        return nm.value, assigned

    def FuseCandidates(self, ext, kf, points, th, invLevelSigma2=None):
        """Candidate search of both ORBmatcher::Fuse overloads (src/ORBmatcher.cc:1325 with the chi-square gate when invLevelSigma2 is
        given, :1543 without).  Returns (bestIdx[M], bestDist[M]); -1 where the reference would not fuse."""
        M = points.view.M
        bi = np.full(M, -1, np.int32); bd = np.full(M, -1, np.int32)
        s2 = None if invLevelSigma2 is None else np.ascontiguousarray(invLevelSigma2, np.float32)
        ext._lib.check(ext._lib.L.orbm_fuse_candidates(ext._h, kf.ref(), points.ref(), float(th), int(s2 is not None),
                                                     None if s2 is None else s2.ctypes.data, bi.ctypes.data, bd.ctypes.data))
        return bi, bd

    def SearchBySim3(self, ext, kf1, kf2, p1in2, p2in1, th):
        """ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, S12, th), src/ORBmatcher.cc:1689.  Returns (nFound, matches12[N1])."""
        m12 = np.full(kf1.view.N, -1, np.int32); nf = C.c_int()
        ext._lib.check(ext._lib.L.orbm_search_by_sim3(ext._h, kf1.ref(), kf2.ref(), p1in2.ref(), p2in1.ref(), float(th), m12.ctypes.data, C.byref(nf)))
        return nf.value, m12

    @staticmethod
    def DescriptorDistance(ext, a, b):
        """All-pairs Hamming distance matrix [len(a), len(b)] of 32-byte descriptors, computed on `ext`'s GPU."""
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32); b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        out = np.zeros((len(a), len(b)), np.int32)
        ext._lib.check(ext._lib.L.orbm_hamming_matrix(ext._h, a.ctypes.data, len(a), b.ctypes.data, len(b), out.ctypes.data))
        return out


class ResidentKeyFrame:
    """orbm_keyframe: the parts of a key frame / frame the vocabulary-bucket searches read (keys, descriptors, mvuRight, mFeatVec), uploaded
    once from a views.key_frame_view.  Usable with every extractor handle of the same device."""

    def __init__(self, ext, kf_view):
        self._lib = ext._lib; self.N = kf_view.view.N
        h = C.c_void_p()
        self._lib.check(self._lib.L.orbm_keyframe_create(ext._h, kf_view.ref(), C.byref(h)))
        self._kf = h

    def close(self):
        if self._kf:
            self._lib.L.orbm_keyframe_destroy(self._kf); self._kf = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ComputeStereoMatches(left, right, bf, b, left_first=0, right_first=0, B=None):
    """Frame::ComputeStereoMatches on the last batches of two extractors (or two halves of one).
    Returns (mvuRight [B,cap], mvDepth [B,cap], n_matches [B]); rows beyond each image's N are -1."""
    B = B or min(left._B - left_first, right._B - right_first)
    L = left._lib
    L.check(L.L.orbm_stereo_match(left._h, left_first, right._h, right_first, B, float(bf), float(b)))
    cap = left.max_keypoints()
    u = np.zeros((B, cap), np.float32); d = np.zeros((B, cap), np.float32); n = np.zeros(B, np.int32)
    L.check(L.L.orbm_stereo_fetch(left._h, B, u.ctypes.data, d.ctypes.data, cap, n.ctypes.data))
    return u, d, n


def StereoFishEyeKnn(left, right, left_first=0, right_first=0, B=None):
    """BFMatcher(NORM_HAMMING).knnMatch(k=2) of left[monoLeft:] vs right[monoRight:] + Lowe ratio 0.7.
    Returns dict of [B,cap] arrays idx0, dist0, idx1, dist1, ratio_ok (rows beyond the query count are -1/0)."""
    B = B or min(left._B - left_first, right._B - right_first)
    L = left._lib
    L.check(L.L.orbm_knn2(left._h, left_first, right._h, right_first, B))
    cap = left.max_keypoints()
    out = {k: np.zeros((B, cap), np.int32) for k in ("idx0", "dist0", "idx1", "dist1")}
    out["ratio_ok"] = np.zeros((B, cap), np.uint8)
    L.check(L.L.orbm_knn2_fetch(left._h, B, out["idx0"].ctypes.data, out["dist0"].ctypes.data, out["idx1"].ctypes.data,
                                out["dist1"].ctypes.data, out["ratio_ok"].ctypes.data, cap))
    return out


class KB8Stereo(C.Structure):
    """OrbmKB8Stereo (include/orbx.h): the two cameras' Kannala-Brandt parameters, Frame::mRlr (row-major), mtlr"""
    _fields_ = [("cam1", C.c_float * 8), ("cam2", C.c_float * 8), ("R12", C.c_float * 9), ("t12", C.c_float * 3)]


def ComputeStereoFishEyeMatches(left, right, cam1, cam2, R12, t12, left_first=0, right_first=0, B=None):
    """Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1530-1587) for B fisheye pairs extracted with the cameras' lapping areas: 2-NN + ratio
    test, then KannalaBrandt8::TriangulateMatches on the device.  cam1 / cam2 = the 8 Kannala-Brandt parameters, (R12, t12) = mRlr, mtlr.
    Returns dict(l2r [B,cap], r2l [B,cap], depth [B,cap], p3d [B,cap,3], n [B])."""
    B = B or min(left._B - left_first, right._B - right_first)
    c = KB8Stereo()
    c.cam1[:] = [float(v) for v in cam1]; c.cam2[:] = [float(v) for v in cam2]
    c.R12[:] = [float(v) for v in np.asarray(R12, np.float32).ravel()]; c.t12[:] = [float(v) for v in np.asarray(t12, np.float32).ravel()]
    L = left._lib
    L.check(L.L.orbm_stereo_fisheye(left._h, left_first, right._h, right_first, B, C.byref(c)))
    cap = left.max_keypoints()
    out = dict(l2r=np.zeros((B, cap), np.int32), r2l=np.zeros((B, cap), np.int32), depth=np.zeros((B, cap), np.float32),
               p3d=np.zeros((B, cap, 3), np.float32), n=np.zeros(B, np.int32))
    L.check(L.L.orbm_stereo_fisheye_fetch(left._h, B, out["l2r"].ctypes.data, out["r2l"].ctypes.data, out["depth"].ctypes.data, out["p3d"].ctypes.data,
                                          out["n"].ctypes.data, cap))
    return out


class _FrustumView(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("qcw", C.c_float * 4), ("camera_type", C.c_int), ("cam", C.c_float * 8),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float), ("mbf", C.c_float), ("log_scale_factor", C.c_float),
                ("nlevels", C.c_int), ("scale_factors", C.c_void_p)]


class _WorldPointView(C.Structure):
    _fields_ = [("M", C.c_int), ("pos", C.c_void_p), ("normal", C.c_void_p), ("min_distance", C.c_void_p), ("max_distance", C.c_void_p), ("is_bad", C.c_void_p),
                ("has_obs", C.c_void_p), ("desc", C.c_void_p)]


class _TrackOut(C.Structure):
    _fields_ = [("in_view", C.c_void_p), ("proj_x", C.c_void_p), ("proj_y", C.c_void_p), ("proj_xr", C.c_void_p), ("depth", C.c_void_p), ("view_cos", C.c_void_p),
                ("scale_level", C.c_void_p)]


class _FrustumRigView(C.Structure):
    _fields_ = [("left", _FrustumView), ("Rrl", C.c_float * 9), ("trl", C.c_float * 3), ("tlr", C.c_float * 3), ("Rwc", C.c_float * 9), ("camera2_type", C.c_int),
                ("cam2", C.c_float * 8)]


class _TrackOutRight(C.Structure):
    _fields_ = [("in_view_r", C.c_void_p), ("proj_xr", C.c_void_p), ("proj_yr", C.c_void_p), ("depth_r", C.c_void_p), ("view_cos_r", C.c_void_p), ("scale_level_r", C.c_void_p)]


def SearchLocalPointsRig(ext, frame2, pose, cam1, cam2, bounds, scale_factors, pos, normal, min_distance, max_distance, is_bad=None, has_obs=None, desc=None,
                         viewing_cos_limit=0.5, th=1.0, far_points=False, th_far=50.0, nnratio=0.8, search=True):
    """Tracking::SearchLocalPoints for a two-camera (fisheye rig) frame: Frame::isInFrustum with Nleft != -1 (src/Frame.cc:754-766 ->
    isInFrustumChecks :1592-1650 per camera) on the device and, with search=True, ORBmatcher::SearchByProjection including its right-camera
    branch (src/ORBmatcher.cc:45-239).  pose = dict(Rcw, tcw, Ow, Rwc, Rrl, trl, tlr) exactly as the Frame holds them; frame2 =
    views.fisheye_frame_view(...) (None with search=False).  Returns (left dict, right dict, assigned [Nleft + Nright] or None, nmatches)."""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    pos, normal, mn, mx, sf = f32(pos).reshape(-1, 3), f32(normal).reshape(-1, 3), f32(min_distance), f32(max_distance), f32(scale_factors)
    M = len(pos)
    V = _FrustumRigView()
    frustum_view(SE3f(), None, cam1, bounds, 0.0, sf, into=V.left)
    V.left.Rcw[:] = f32(pose["Rcw"]).ravel().tolist(); V.left.tcw[:] = f32(pose["tcw"]).tolist()       # the Frame's own members, untouched
    V.left.Ow[:] = [float(v) for v in f32(pose["Ow"])]
    V.left.scale_factors = sf.ctypes.data
    V.Rrl[:] = f32(pose["Rrl"]).ravel().tolist(); V.trl[:] = f32(pose["trl"]).tolist(); V.tlr[:] = f32(pose["tlr"]).tolist(); V.Rwc[:] = f32(pose["Rwc"]).ravel().tolist()
    cam2 = [float(v) for v in cam2]
    V.camera2_type = 1 if len(cam2) == 8 else 0
    V.cam2[:] = cam2 + [0.0] * (8 - len(cam2))
    P = _WorldPointView()
    bad = None if is_bad is None else np.ascontiguousarray(is_bad, np.uint8); obs = None if has_obs is None else np.ascontiguousarray(has_obs, np.uint8)
    d = None if desc is None else np.ascontiguousarray(desc, np.uint8)
    P.M = M; P.pos = pos.ctypes.data; P.normal = normal.ctypes.data; P.min_distance = mn.ctypes.data; P.max_distance = mx.ctypes.data
    P.is_bad = None if bad is None else bad.ctypes.data; P.has_obs = None if obs is None else obs.ctypes.data; P.desc = None if d is None else d.ctypes.data
    M1 = max(M, 1)
    tl = dict(in_view=np.zeros(M1, np.uint8), proj_x=np.zeros(M1, np.float32), proj_y=np.zeros(M1, np.float32), proj_xr=np.zeros(M1, np.float32), depth=np.zeros(M1, np.float32),
              view_cos=np.zeros(M1, np.float32), scale_level=np.zeros(M1, np.int32))
    tr = dict(in_view_r=np.zeros(M1, np.uint8), proj_xr=np.zeros(M1, np.float32), proj_yr=np.zeros(M1, np.float32), depth_r=np.zeros(M1, np.float32),
              view_cos_r=np.zeros(M1, np.float32), scale_level_r=np.zeros(M1, np.int32))
    TL = _TrackOut(*[tl[k].ctypes.data for k in ("in_view", "proj_x", "proj_y", "proj_xr", "depth", "view_cos", "scale_level")])
    TR = _TrackOutRight(*[tr[k].ctypes.data for k in ("in_view_r", "proj_xr", "proj_yr", "depth_r", "view_cos_r", "scale_level_r")])
    L = ext._lib
    cut = lambda dct: {k: v[:M] for k, v in dct.items()}
    if not search:
        L.check(L.L.orbm_is_in_frustum_rig(ext._h, C.byref(V), C.byref(P), float(viewing_cos_limit), C.byref(TL), C.byref(TR)))
        return cut(tl), cut(tr), None, 0
    N = frame2.view.left.N + frame2.view.right.N
    assigned = np.full(max(N, 1), -1, np.int32); n = C.c_int(0)
    L.check(L.L.orbm_search_local_points_fisheye(ext._h, frame2.ref(), C.byref(V), C.byref(P), float(viewing_cos_limit), float(th), int(far_points), float(th_far), float(nnratio),
                                                 C.byref(TL), C.byref(TR), assigned.ctypes.data, C.byref(n)))
    return cut(tl), cut(tr), assigned[:N], n.value


class _Projection(C.Structure):
    _fields_ = [("q", C.c_float * 4), ("t", C.c_float * 3), ("second", C.c_int), ("q2", C.c_float * 4), ("t2", C.c_float * 3), ("s2", C.c_float), ("Ow", C.c_float * 3),
                ("dist_mode", C.c_int), ("depth_test", C.c_int), ("camera_type", C.c_int), ("cam", C.c_float * 8), ("inline_pinhole", C.c_int), ("min_x", C.c_float),
                ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float), ("bounds_mode", C.c_int), ("distance_test", C.c_int), ("angle_test", C.c_int), ("bf", C.c_float)]


class _ProjectIn(C.Structure):
    _fields_ = [("M", C.c_int), ("pos", C.c_void_p), ("normal", C.c_void_p), ("min_inv", C.c_void_p), ("max_inv", C.c_void_p), ("skip", C.c_void_p)]


class _ProjectOut(C.Structure):
    _fields_ = [("valid", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("ur", C.c_void_p), ("inv_z", C.c_void_p), ("dist", C.c_void_p)]


def ProjectPoints(ext, pose, cam, bounds, pos, normal=None, min_inv=None, max_inv=None, skip=None, Ow=None, second=None, depth_test=1, bounds_mode=0, inline_pinhole=False,
                  dist_mode=0, angle_test=False, bf=0.0):
    """orbm_project_points: the geometry in front of the projection-type searches (transform, depth test, projection, image test, distance
    range, viewing angle) for M map points on the device.  pose = sophus.SE3f (or (R, t), taken through the SE3(R, t) constructor): the device
    evaluates `pose * p` as Sophus does, on the unit quaternion.  second = a sophus.Sim3f (SearchBySim3's S21 / S12) or SE3f (the rig's Trl)
    applied behind the pose.  Returns dict(valid, u, v, ur, inv_z, dist)."""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    pos = f32(pos).reshape(-1, 3); M = len(pos)
    S = _Projection()
    T = as_se3(pose)
    S.q[:] = [float(v) for v in T.unit_quaternion()]; S.t[:] = [float(v) for v in T.translation()]
    if isinstance(second, Sim3f):
        S.second = 1; S.q2[:] = [float(v) for v in second.quaternion()]; S.t2[:] = [float(v) for v in second.translation()]; S.s2 = float(second.scale())
    elif second is not None:
        X = as_se3(second)
        S.second = 2; S.q2[:] = [float(v) for v in X.unit_quaternion()]; S.t2[:] = [float(v) for v in X.translation()]; S.s2 = 1.0
    if Ow is not None:
        S.Ow[:] = f32(Ow).tolist()
    cam = [float(v) for v in cam]
    S.camera_type = 1 if len(cam) == 8 else 0; S.cam[:] = cam + [0.0] * (8 - len(cam)); S.inline_pinhole = int(inline_pinhole)
    S.min_x, S.max_x, S.min_y, S.max_y = [float(v) for v in bounds]
    S.dist_mode, S.depth_test, S.bounds_mode, S.angle_test, S.bf = int(dist_mode), int(depth_test), int(bounds_mode), int(angle_test), float(bf)
    S.distance_test = int(min_inv is not None and max_inv is not None)
    keep = [pos] + [None if a is None else (f32(a) if i < 3 else np.ascontiguousarray(a, np.uint8)) for i, a in enumerate((normal, min_inv, max_inv, skip))]
    ptr = lambda a: None if a is None else a.ctypes.data
    I = _ProjectIn(M, ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), ptr(keep[4]))
    M1 = max(M, 1)
    out = dict(valid=np.zeros(M1, np.uint8), u=np.zeros(M1, np.float32), v=np.zeros(M1, np.float32), ur=np.zeros(M1, np.float32), inv_z=np.zeros(M1, np.float32),
               dist=np.zeros(M1, np.float32))
    O = _ProjectOut(*[out[k].ctypes.data for k in ("valid", "u", "v", "ur", "inv_z", "dist")])
    ext._lib.check(ext._lib.L.orbm_project_points(ext._h, C.byref(S), C.byref(I), C.byref(O)))
    return {k: v[:M] for k, v in out.items()}


class _LastFrameBatch(C.Structure):
    _fields_ = [("cap_last", C.c_int), ("n", C.c_void_p), ("pos", C.c_void_p), ("valid", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p), ("has_obs", C.c_void_p),
                ("desc", C.c_void_p)]


class LastFrameBatch:
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) for a batch of frames on the device
    (orbm_search_by_projection_lastframe_batch): current frames = images [first, first + B) of ext's last extraction; per frame the map points
    of its last frame.  enqueue() is asynchronous, fetch() returns (assigned [B, cap], nmatches [B])."""

    def __init__(self, ext, B, cam, bounds, mbf, scale_factors):
        self.ext, self.B = ext, B
        self.cam, self.bounds, self.mbf = cam, bounds, mbf
        self.sf = np.ascontiguousarray(scale_factors, np.float32)
        self.views = (_FrustumView * B)()
        self.cap = ext.max_keypoints()
        self.assigned = np.full((B, self.cap), -1, np.int32); self.nm = np.zeros(B, np.int32)

    def set_poses(self, poses):
        for b, pose in enumerate(poses):
            frustum_view(as_se3(pose), None, self.cam, self.bounds, self.mbf, self.sf, into=self.views[b])
            self.views[b].scale_factors = self.sf.ctypes.data

    def enqueue(self, n, pos, valid, octave, angle, has_obs, desc, th, forward=None, backward=None, check_orientation=True, occupied=None, use_u_right=True, first=0):
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        u8 = lambda a: None if a is None else np.ascontiguousarray(a, np.uint8)
        self._again = lambda: self.enqueue(n, pos, valid, octave, angle, has_obs, desc, th, forward, backward, check_orientation, occupied, use_u_right, first)
        pos = f32(pos); capL = pos.shape[1]
        self._keep = (np.ascontiguousarray(n, np.int32), pos, u8(valid), np.ascontiguousarray(octave, np.int32), f32(angle), u8(has_obs), u8(desc), u8(forward), u8(backward), u8(occupied))
        k = self._keep
        ptr = lambda a: None if a is None else a.ctypes.data
        lb = _LastFrameBatch(capL, ptr(k[0]), ptr(k[1]), ptr(k[2]), ptr(k[3]), ptr(k[4]), ptr(k[5]), ptr(k[6]))
        L = self.ext._lib
        L.check(L.L.orbm_search_by_projection_lastframe_batch(self.ext._h, int(first), self.B, self.views, C.byref(lb), float(th), ptr(k[7]), ptr(k[8]), int(bool(check_orientation)),
                                                              ptr(k[9]), int(bool(use_u_right))))

    def fetch(self):
        L = self.ext._lib
        rc = L.L.orbm_search_local_points_fetch(self.ext._h, self.assigned.ctypes.data, self.cap, self.nm.ctypes.data, None)
        if rc == -4:                 # ORBX_E_CAPACITY: pool enlarged, run the batch again
            self._again()
            rc = L.L.orbm_search_local_points_fetch(self.ext._h, self.assigned.ctypes.data, self.cap, self.nm.ctypes.data, None)
        L.check(rc)
        return self.assigned, self.nm


class _KeyFramePointBatch(C.Structure):
    _fields_ = [("cap_kf", C.c_int), ("n", C.c_void_p), ("pos", C.c_void_p), ("valid", C.c_void_p), ("min_distance", C.c_void_p), ("max_distance", C.c_void_p),
                ("angle", C.c_void_p), ("desc", C.c_void_p)]


class KeyFrameBatch(LastFrameBatch):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) - relocalisation - for a batch of frames on the device
    (orbm_search_by_projection_keyframe_batch): current frames = images [first, first + B) of ext's last extraction, per frame the map points of
    its candidate key frame.  set_poses() as LastFrameBatch; enqueue() is asynchronous, fetch() returns (assigned [B, cap], nmatches [B])."""

    def enqueue(self, n, pos, valid, min_distance, max_distance, angle, desc, th, orb_dist, check_orientation=True, occupied=None, first=0):
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        u8 = lambda a: None if a is None else np.ascontiguousarray(a, np.uint8)
        self._again = lambda: self.enqueue(n, pos, valid, min_distance, max_distance, angle, desc, th, orb_dist, check_orientation, occupied, first)
        pos = f32(pos); capK = pos.shape[1]
        self._keep = (np.ascontiguousarray(n, np.int32), pos, u8(valid), f32(min_distance), f32(max_distance), f32(angle), u8(desc), u8(occupied))
        k = self._keep
        ptr = lambda a: None if a is None else a.ctypes.data
        kb = _KeyFramePointBatch(capK, ptr(k[0]), ptr(k[1]), ptr(k[2]), ptr(k[3]), ptr(k[4]), ptr(k[5]), ptr(k[6]))
        L = self.ext._lib
        L.check(L.L.orbm_search_by_projection_keyframe_batch(self.ext._h, int(first), self.B, self.views, C.byref(kb), float(th), int(orb_dist), int(bool(check_orientation)), ptr(k[7])))


class ResidentPoints:
    """orbm_points: position, normal, distance limits and descriptor of a set of map points, uploaded once (the local map)."""

    def __init__(self, ext, pos, normal, min_distance, max_distance, desc):
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        self._lib = ext._lib
        pos, normal, mn, mx, d = f32(pos).reshape(-1, 3), f32(normal).reshape(-1, 3), f32(min_distance), f32(max_distance), np.ascontiguousarray(desc, np.uint8)
        P = _WorldPointView(); P.M = len(pos)
        P.pos, P.normal, P.min_distance, P.max_distance, P.desc = pos.ctypes.data, normal.ctypes.data, mn.ctypes.data, mx.ctypes.data, d.ctypes.data
        h = C.c_void_p()
        self._lib.check(self._lib.L.orbm_points_create(ext._h, C.byref(P), C.byref(h)))
        self._p, self.M = h, len(pos)

    def close(self):
        if self._p:
            self._lib.L.orbm_points_destroy(self._p); self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frustum_view(Rcw, tcw, cam, bounds, mbf, scale_factors, into=None):
    """OrbmFrustumView of one frame.  The pose is a sophus.SE3f (pass it as Rcw, tcw = None) or (Rcw, tcw), which enters through Sophus' SE3(R, t)
    constructor as in Frame::SetPose(Sophus::SE3f(R, t)); the view then holds what Frame::UpdatePoseMatrices derives (src/Frame.cc:594-598):
    mRcw = mTcw.rotationMatrix(), mtcw, mOw = mTcw.inverse().translation(), plus the unit quaternion for the searches that evaluate Tcw * p.
    Camera (4 pinhole or 8 Kannala-Brandt parameters), image bounds (min_x, max_x, min_y, max_y), mbf, the scale factors.
    Returns (view, the float32 scale-factor array the view points into - keep it alive)."""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    T = Rcw if isinstance(Rcw, SE3f) else SE3f(Rcw, tcw)
    sf = f32(scale_factors)
    V = into if into is not None else _FrustumView()
    V.Rcw[:] = [float(v) for v in T.rotationMatrix().ravel()]; V.tcw[:] = [float(v) for v in T.translation()]
    V.Ow[:] = [float(v) for v in T.inverse().translation()]
    V.qcw[:] = [float(v) for v in T.unit_quaternion()]
    cam = [float(v) for v in cam]
    V.camera_type = 1 if len(cam) == 8 else 0
    V.cam[:] = cam + [0.0] * (8 - len(cam))
    V.min_x, V.max_x, V.min_y, V.max_y = [float(v) for v in bounds]
    V.mbf = float(mbf); V.log_scale_factor = float(np.float32(np.log(np.float64(sf[1])))) if len(sf) > 1 else 1.0
    V.nlevels = len(sf); V.scale_factors = sf.ctypes.data
    return V, sf


def ComputeStereoFromRGBD(ext, depth, mbf, first=0, device_ptr=None, shape=None):
    """Frame::ComputeStereoFromRGBD (src/Frame.cc:1361-1391) for the frames [first, first + B) of ext's last batch: depth [B, H, W] float32 (CV_32F,
    already multiplied by the depth map factor), host array or (device_ptr, shape).  Asynchronous; fetch with StereoFetch(ext, B) or keep on the
    device for SearchLocalPointsBatch."""
    L = ext._lib
    if device_ptr is None:
        depth = np.ascontiguousarray(depth, np.float32)
        B, H, W = depth.shape
        ext._keep_depth = depth
        L.check(L.L.orbm_stereo_from_depth(ext._h, int(first), B, depth.ctypes.data, W, H * W, 0, float(mbf)))
    else:
        B, H, W = shape
        L.check(L.L.orbm_stereo_from_depth(ext._h, int(first), B, device_ptr, W, H * W, 1, float(mbf)))
    return B


def StereoFetch(ext, B):
    """(mvuRight [B, cap], mvDepth [B, cap], matches [B]) of the last orbm_stereo_match / orbm_stereo_from_depth of `ext`."""
    cap = ext.max_keypoints()
    u = np.zeros((B, cap), np.float32); d = np.zeros((B, cap), np.float32); n = np.zeros(B, np.int32)
    ext._lib.check(ext._lib.L.orbm_stereo_fetch(ext._h, B, u.ctypes.data, d.ctypes.data, cap, n.ctypes.data))
    return u, d, n


class LocalPointsBatch:
    """Tracking::SearchLocalPoints for a batch of frames that stay on the device (orbm_search_local_points_batch): the frames are images
    [first, first + B) of ext's last extraction, the local map is a ResidentPoints, poses = B x (Rcw, tcw).  enqueue() is asynchronous, fetch()
    returns (assigned [B, cap], nmatches [B], in_view [B, M] or None)."""

    def __init__(self, ext, resident, B, cam, bounds, mbf, scale_factors):
        self.ext, self.res, self.B = ext, resident, B
        self.cam, self.bounds, self.mbf = cam, bounds, mbf
        self.sf = np.ascontiguousarray(scale_factors, np.float32)
        self.views = (_FrustumView * B)()
        self.cap = ext.max_keypoints()
        self.assigned = np.full((B, self.cap), -1, np.int32); self.nm = np.zeros(B, np.int32)
        self.in_view = np.zeros((B, max(resident.M, 1)), np.uint8)

    def set_poses(self, poses):
        for b, pose in enumerate(poses):
            frustum_view(as_se3(pose), None, self.cam, self.bounds, self.mbf, self.sf, into=self.views[b])
            self.views[b].scale_factors = self.sf.ctypes.data

    def enqueue(self, first=0, is_bad=None, has_obs=None, occupied=None, use_u_right=True, viewing_cos_limit=0.5, th=1.0, far_points=False, th_far=50.0, nnratio=0.8,
                want_in_view=False):
        L = self.ext._lib
        u8 = lambda a: None if a is None else np.ascontiguousarray(a, np.uint8)
        self._again = lambda: self.enqueue(first, is_bad, has_obs, occupied, use_u_right, viewing_cos_limit, th, far_points, th_far, nnratio, want_in_view)
        self._keep = (u8(is_bad), u8(has_obs), u8(occupied))
        ptr = lambda a: None if a is None else a.ctypes.data
        self._want = bool(want_in_view)
        L.check(L.L.orbm_search_local_points_batch(self.ext._h, int(first), self.B, self.views, self.res._p, ptr(self._keep[0]), ptr(self._keep[1]), ptr(self._keep[2]),
                                                   int(bool(use_u_right)), float(viewing_cos_limit), float(th), int(far_points), float(th_far), float(nnratio), int(self._want)))

    def fetch(self):
        L = self.ext._lib
        rc = L.L.orbm_search_local_points_fetch(self.ext._h, self.assigned.ctypes.data, self.cap, self.nm.ctypes.data, self.in_view.ctypes.data if self._want else None)
        if rc == -4:                 # ORBX_E_CAPACITY: the candidate pool was too small for this scene and has been enlarged - run the batch again
            self._again()
            rc = L.L.orbm_search_local_points_fetch(self.ext._h, self.assigned.ctypes.data, self.cap, self.nm.ctypes.data, self.in_view.ctypes.data if self._want else None)
        L.check(rc)
        return self.assigned, self.nm, (self.in_view if self._want else None)


def SearchLocalPoints(ext, frame, Rcw, tcw, cam, bounds, mbf, scale_factors, pos, normal, min_distance, max_distance, is_bad=None, has_obs=None, desc=None,
                      viewing_cos_limit=0.5, th=1.0, far_points=False, th_far=50.0, nnratio=0.8, search=True, prepared=False, resident=None):
    """Frame::isInFrustum (src/Frame.cc:667-773) for M map points and, with search=True, ORBmatcher::SearchByProjection(F, points, th, ...)
    (src/ORBmatcher.cc:45-167) on those in view - Tracking::SearchLocalPoints (src/Tracking.cc:4009-4067) on the device.
    cam: (fx, fy, cx, cy) or the 8 Kannala-Brandt parameters; bounds = (min_x, max_x, min_y, max_y); frame: views.frame_view(...).
    Returns (track dict, assigned[N] or None, nmatches)."""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    pos, normal, mn, mx = f32(pos).reshape(-1, 3), f32(normal).reshape(-1, 3), f32(min_distance), f32(max_distance)
    M = len(pos)
    V, sf = frustum_view(Rcw, tcw, cam, bounds, mbf, scale_factors)
    P = _WorldPointView()
    bad = None if is_bad is None else np.ascontiguousarray(is_bad, np.uint8); obs = None if has_obs is None else np.ascontiguousarray(has_obs, np.uint8)
    d = None if desc is None else np.ascontiguousarray(desc, np.uint8)
    P.M = M; P.pos = pos.ctypes.data; P.normal = normal.ctypes.data; P.min_distance = mn.ctypes.data; P.max_distance = mx.ctypes.data
    P.is_bad = None if bad is None else bad.ctypes.data; P.has_obs = None if obs is None else obs.ctypes.data; P.desc = None if d is None else d.ctypes.data
    tr = dict(in_view=np.zeros(max(M, 1), np.uint8), proj_x=np.zeros(max(M, 1), np.float32), proj_y=np.zeros(max(M, 1), np.float32), proj_xr=np.zeros(max(M, 1), np.float32),
              depth=np.zeros(max(M, 1), np.float32), view_cos=np.zeros(max(M, 1), np.float32), scale_level=np.zeros(max(M, 1), np.int32))
    T = _TrackOut(*[tr[k].ctypes.data for k in ("in_view", "proj_x", "proj_y", "proj_xr", "depth", "view_cos", "scale_level")])
    L = ext._lib
    if not search:
        L.check(L.L.orbm_is_in_frustum(ext._h, C.byref(V), C.byref(P), float(viewing_cos_limit), C.byref(T)))
        return {k: v[:M] for k, v in tr.items()}, None, 0
    assigned = np.full(max(frame.view.N, 1), -1, np.int32); n = C.c_int(0)
    keep = (V, P, T, sf, pos, normal, mn, mx, bad, obs, d, tr)

    def call(_keep=keep):
        """the C ABI call alone (arguments prebuilt): what a C++ caller pays.  resident: a ResidentPoints made from the same points"""
        if resident is not None:
            return L.L.orbm_search_local_points_resident(ext._h, frame.ref(), C.byref(V), resident._p, P.is_bad, P.has_obs, float(viewing_cos_limit), float(th), int(far_points),
                                                         float(th_far), float(nnratio), C.byref(T), assigned.ctypes.data, C.byref(n))
        return L.L.orbm_search_local_points(ext._h, frame.ref(), C.byref(V), C.byref(P), float(viewing_cos_limit), float(th), int(far_points), float(th_far), float(nnratio),
                                            C.byref(T), assigned.ctypes.data, C.byref(n))
    if prepared:
        return call
    L.check(call())
    return {k: v[:M] for k, v in tr.items()}, assigned[:frame.view.N], n.value


def GetFeaturesInArea(ext, frame, x, y, r, minLevel=-1, maxLevel=-1):
    """Frame::GetFeaturesInArea (src/Frame.cc:859): keypoint indices in the reference's order."""
    cap = max(frame.view.N, 1)
    out = np.zeros(cap, np.int32)
    n = ext._lib.L.orbm_get_features_in_area(ext._h, frame.ref(), float(x), float(y), float(r), int(minLevel), int(maxLevel), out.ctypes.data, cap)
    if n < 0:
        ext._lib.check(n)
    return out[:n].copy()


def AreaSearchBatch(ext, frame, queries, query_desc):
    """Batched GetFeaturesInArea + Hamming distances (orbm_area_search_batch).  queries: [Q,5] float array (x, y, r, minLevel,
    maxLevel); returns per query a list of (idx, dist, level) in the reference's GetFeaturesInArea order."""
    q = np.zeros(len(queries), np.dtype([("x", "<f4"), ("y", "<f4"), ("r", "<f4"), ("mn", "<i4"), ("mx", "<i4")]))
    qa = np.asarray(queries, np.float64).reshape(-1, 5)
    q["x"], q["y"], q["r"], q["mn"], q["mx"] = qa[:, 0], qa[:, 1], qa[:, 2], qa[:, 3].astype(np.int32), qa[:, 4].astype(np.int32)
    d = np.ascontiguousarray(query_desc, np.uint8).reshape(len(q), 32)
    Q = len(q)
    start = np.zeros(Q, np.int32); count = np.zeros(Q, np.int32)
    cap = max(64 * Q, 1024)
    for _ in range(2):
        idx = np.zeros(cap, np.int32); dist = np.zeros(cap, np.int32); lvl = np.zeros(cap, np.int32)
        tot = ext._lib.L.orbm_area_search_batch(ext._h, frame.ref(), q.ctypes.data, d.ctypes.data, Q, start.ctypes.data, count.ctypes.data,
                                                idx.ctypes.data, dist.ctypes.data, lvl.ctypes.data, cap)
        if tot < 0:
            ext._lib.check(tot)
        if tot <= cap:
            break
        cap = tot
    return [list(zip(idx[s:s + c].tolist(), dist[s:s + c].tolist(), lvl[s:s + c].tolist())) for s, c in zip(start.tolist(), count.tolist())]


def ComputeDistinctiveDescriptors(ext, desc, start):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:438) for many map points: desc [total, 32] uint8, start [P+1] offsets.
    Returns best[P]: which of each point's descriptors becomes mDescriptor (-1: point without descriptors)."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); st = np.ascontiguousarray(start, np.int32)
    P = len(st) - 1
    best = np.full(max(P, 1), -1, np.int32)
    ext._lib.check(ext._lib.L.orbm_distinctive_descriptors(ext._h, d.ctypes.data if len(d) else None, st.ctypes.data, P, best.ctypes.data))
    return best[:P]
