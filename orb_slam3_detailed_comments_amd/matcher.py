"""Host-side mirror of the reference's descriptor matchers on top of the HIP library:
``ORBmatcher::DescriptorDistance`` (src/ORBmatcher.cc:2383-2403), ``Frame::ComputeStereoMatches``
(src/Frame.cc:1102-1358) and the kNN + ratio part of ``Frame::ComputeStereoFishEyeMatches`` (:1553-1562)."""
import numpy as np


class ORBmatcher:
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30   # src/ORBmatcher.cc:35-37

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)

    @staticmethod
    def DescriptorDistance(ext, a, b):
        """All-pairs Hamming distance matrix [len(a), len(b)] of 32-byte descriptors, computed on `ext`'s GPU."""
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32); b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        out = np.zeros((len(a), len(b)), np.int32)
        ext._lib.check(ext._lib.L.orbm_hamming_matrix(ext._h, a.ctypes.data, len(a), b.ctypes.data, len(b), out.ctypes.data))
        return out


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
