"""ctypes bindings for the CPU oracle (TEST INFRASTRUCTURE: tests/, smoke() and bench cpu_baseline only)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build():
    subprocess.run(["make", "-j8", "-C", ORACLE_DIR, "-s"], check=True, stdout=subprocess.DEVNULL)


def _load(path):
    return C.CDLL(path) if os.path.exists(path) else None


_ORACLE = None
_REF = None


def oracle():
    global _ORACLE
    if _ORACLE is None:
        p = os.path.join(ORACLE_DIR, "liborb_oracle.so")
        if not os.path.exists(p):
            build()
        L = C.CDLL(p)
        L.orbo_create.restype = C.c_void_p
        L.orbo_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orbo_destroy.argtypes = [C.c_void_p]
        L.orbo_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orbo_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orbo_level_info.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 3 + [C.POINTER(C.c_float)]
        for f in (L.orbo_level_image, L.orbo_level_blurred):
            f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orbo_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orbo_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orbo_quadtree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orbo_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orbo_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float,
                                          C.c_void_p, C.c_void_p]
        L.orbo_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        for f in (L.orbo_cosf, L.orbo_sinf):
            f.restype = C.c_float
            f.argtypes = [C.c_float]
        L.orbo_fast_atan2.restype = C.c_float
        L.orbo_fast_atan2.argtypes = [C.c_float, C.c_float]
        _ORACLE = L
    return _ORACLE


def reference(fma=False):
    """The reference's own ORBextractor.cc built against the shim (oracle/_ref); None if absent.  fma = True: the build with the reference's own
    optimisation flags on an FMA host (-O3, AVX2 + FMA, contraction on: oracle/Makefile _ref/libref_orb_fma.so) - not cached, for tests/test_fma_contract.py."""
    global _REF
    if _REF is None or fma:
        p = os.path.join(ORACLE_DIR, "_ref", "libref_orb_fma.so" if fma else "libref_orb.so")
        if not os.path.exists(p) and os.path.exists("/root/reference/src/ORBextractor.cc"):
            build()
        if not os.path.exists(p):
            return None
        L = C.CDLL(p)
        L.ref_orb_create.restype = C.c_void_p
        L.ref_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.ref_orb_destroy.argtypes = [C.c_void_p]
        L.ref_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.ref_orb_pyramid_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_orb_keypoints_per_level.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_orb_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        if fma:
            return L
        _REF = L
    return _REF


class OracleExtractor:
    """Our C++ restatement (oracle/orb_oracle.cpp)."""

    def __init__(self, nfeatures=1200, scale=1.2, nlevels=8, ini=20, mn=7, gauss_variant=0):
        self.L = oracle()
        self.nlevels = nlevels
        self.cap = nfeatures + 16 * nlevels + 64        # (a level returns up to max(quota + 3, 4 * roots) keypoints: 12 per level for a one-feature budget on a wide image)
        self.h = self.L.orbo_create(nfeatures, scale, nlevels, ini, mn, gauss_variant)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orbo_destroy(self.h)
            self.h = None

    def extract(self, img, lap=(0, 0)):
        img = np.ascontiguousarray(img, np.uint8)
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int()
        mono = self.L.orbo_extract(self.h, img.ctypes.data, img.shape[1], img.shape[0], img.strides[0],
                                   lap[0], lap[1], kps.ctypes.data, desc.ctypes.data, self.cap, C.byref(n))
        assert mono > -2, "oracle extract failed %d" % mono
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def level_info(self, l):
        w, h, q, s = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        self.L.orbo_level_info(self.h, l, C.byref(w), C.byref(h), C.byref(q), C.byref(s))
        return w.value, h.value, q.value, s.value

    def level_image(self, l, blurred=False):
        w, h, _, _ = self.level_info(l)
        a = np.zeros((h, w), np.uint8)
        (self.L.orbo_level_blurred if blurred else self.L.orbo_level_image)(self.h, l, a.ctypes.data)
        return a

    def level_candidates(self, l):
        cap = 1 << 18
        a = np.zeros((cap, 3), np.int32)
        n = self.L.orbo_level_candidates(self.h, l, a.ctypes.data, cap)
        if n > cap:                                   # (17-Mpixel noise: 2.6 M candidates on level 0)
            a = np.zeros((n, 3), np.int32)
            assert self.L.orbo_level_candidates(self.h, l, a.ctypes.data, n) == n
        return a[:n].copy()

    def level_keypoints(self, l):
        a = np.zeros(self.cap, KP_DTYPE)
        n = self.L.orbo_level_keypoints(self.h, l, a.ctypes.data, self.cap)
        return a[:n].copy()

    def tables(self):
        nl = self.nlevels
        q = np.zeros(nl, np.int32); um = np.zeros(16, np.int32)
        f = [np.zeros(nl, np.float32) for _ in range(4)]
        self.L.orbo_tables(self.h, q.ctypes.data, um.ctypes.data, *[x.ctypes.data for x in f])
        return q, um, f


class ReferenceExtractor:
    """The reference's own ORBextractor (src/ORBextractor.cc) built against the OpenCV shim."""

    def __init__(self, nfeatures=1200, scale=1.2, nlevels=8, ini=20, mn=7, gauss_variant=0, fma=False):
        self.L = reference(fma)
        assert self.L is not None, "oracle/_ref/libref_orb.so missing"
        self.nlevels = nlevels
        self.gv = gauss_variant
        self.cap = nfeatures + 16 * nlevels + 64        # (a level returns up to max(quota + 3, 4 * roots) keypoints: 12 per level for a one-feature budget on a wide image)
        self.h = self.L.ref_orb_create(nfeatures, scale, nlevels, ini, mn)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_orb_destroy(self.h)
            self.h = None

    def extract(self, img, lap=(0, 0)):
        img = np.ascontiguousarray(img, np.uint8)
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int()
        mono = self.L.ref_orb_extract(self.h, img.ctypes.data, img.shape[1], img.shape[0], img.strides[0],
                                      lap[0], lap[1], self.gv, kps.ctypes.data, desc.ctypes.data, self.cap, C.byref(n))
        assert n.value <= self.cap
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def level_image(self, l):
        cap = 1 << 22
        a = np.zeros(cap, np.uint8)
        w, h = C.c_int(), C.c_int()
        r = self.L.ref_orb_pyramid_level(self.h, l, a.ctypes.data, cap, C.byref(w), C.byref(h))
        assert r == 0
        return a[:w.value * h.value].reshape(h.value, w.value).copy()

    def tables(self):
        nl = self.nlevels
        q = np.zeros(nl, np.int32); um = np.zeros(16, np.int32); pat = np.zeros(1024, np.int32)
        f = [np.zeros(nl, np.float32) for _ in range(4)]
        self.L.ref_orb_tables(self.h, q.ctypes.data, um.ctypes.data, pat.ctypes.data, *[x.ctypes.data for x in f])
        return q, um, pat, f


def kps_equal(a, b):
    """Bit-exact comparison of keypoint records (angle compared as bit pattern)."""
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def oracle_stereo(oL, oR, kL, dL, kR, dR, bf, b):
    """Frame::ComputeStereoMatches restatement on two OracleExtractor objects (their last extract() pyramids)."""
    L = oracle()
    kL = np.ascontiguousarray(kL); kR = np.ascontiguousarray(kR)
    dL = np.ascontiguousarray(dL); dR = np.ascontiguousarray(dR)
    u = np.zeros(len(kL), np.float32); d = np.zeros(len(kL), np.float32)
    n = L.orbo_stereo_matches(oL.h, oR.h, kL.ctypes.data, len(kL), dL.ctypes.data, kR.ctypes.data, len(kR), dR.ctypes.data,
                              float(bf), float(b), u.ctypes.data, d.ctypes.data)
    return u, d, n


def oracle_knn2(q, t):
    L = oracle()
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    nq = len(q)
    out = [np.zeros(nq, np.int32) for _ in range(4)]
    ok = np.zeros(nq, np.uint8)
    L.orbo_knn2(q.ctypes.data, nq, t.ctypes.data, len(t), out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data,
                out[3].ctypes.data, ok.ctypes.data)
    return dict(idx0=out[0], dist0=out[1], idx1=out[2], dist1=out[3], ratio_ok=ok)


def oracle_hamming(a, b):
    L = oracle()
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return L.orbo_descriptor_distance(a.ctypes.data, b.ctypes.data)


# ---- guided searches (M3-M6): oracle restatements on the same views as the product ABI ----
def oracle_features_in_area(frame, x, y, r, minLevel=-1, maxLevel=-1):
    L = oracle()
    L.orbo_get_features_in_area.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]
    out = np.zeros(max(frame.view.N, 1), np.int32)
    n = L.orbo_get_features_in_area(frame.ref(), x, y, r, minLevel, maxLevel, out.ctypes.data, len(out))
    return out[:n].copy()


def oracle_search_by_projection_mappoints(frame, mps, th, bFar, thFar, nnratio):
    L = oracle()
    L.orbo_search_by_projection_mappoints.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p]
    a = np.full(max(frame.view.N, 1), -1, np.int32)
    n = L.orbo_search_by_projection_mappoints(frame.ref(), mps.ref(), th, int(bFar), thFar, nnratio, a.ctypes.data)
    return n, a[:frame.view.N]


def oracle_search_by_projection_frame(cur, last, th, fwd, bwd, check_ori):
    L = oracle()
    L.orbo_search_by_projection_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]
    a = np.full(max(cur.view.N, 1), -1, np.int32)
    n = L.orbo_search_by_projection_frame(cur.ref(), last.ref(), th, int(fwd), int(bwd), int(check_ori), a.ctypes.data)
    return n, a[:cur.view.N]


def oracle_search_for_triangulation(k1, k2, F12, ep, only_stereo, coarse, check_ori):
    L = oracle()
    L.orbo_search_for_triangulation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    F12 = np.ascontiguousarray(F12, np.float32).reshape(9); ep = np.ascontiguousarray(ep, np.float32).reshape(2)
    m = np.full(max(k1.view.N, 1), -1, np.int32)
    n = L.orbo_search_for_triangulation(k1.ref(), k2.ref(), F12.ctypes.data, ep.ctypes.data, int(only_stereo), int(coarse), int(check_ori), m.ctypes.data)
    idx = np.nonzero(m[:k1.view.N] >= 0)[0]
    return n, [(int(i), int(m[i])) for i in idx]


def oracle_search_by_bow(k1, k2, nnratio, frame_version, check_ori):
    L = oracle()
    L.orbo_search_by_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]
    m = np.full(max(k1.view.N, 1), -1, np.int32)
    n = L.orbo_search_by_bow(k1.ref(), k2.ref(), nnratio, int(frame_version), int(check_ori), m.ctypes.data)
    return n, m[:k1.view.N]


def oracle_search_for_initialization(f1, f2, prev, window, nnratio, check_ori):
    L = oracle()
    L.orbo_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    m = np.full(max(f1.view.N, 1), -1, np.int32)
    n = L.orbo_search_for_initialization(f1.ref(), f2.ref(), prev.ctypes.data, int(window), nnratio, int(check_ori), m.ctypes.data)
    return n, m[:f1.view.N]


def oracle_search_by_projection_sim3(kf, pts, th, ratio):
    L = oracle()
    L.orbo_search_by_projection_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    a = np.full(max(kf.view.N, 1), -1, np.int32)
    n = L.orbo_search_by_projection_sim3(kf.ref(), pts.ref(), th, ratio, a.ctypes.data)
    return n, a[:kf.view.N]


def oracle_search_by_projection_keyframe(cur, pts, th, orb_dist, check_ori):
    L = oracle()
    L.orbo_search_by_projection_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]
    a = np.full(max(cur.view.N, 1), -1, np.int32)
    n = L.orbo_search_by_projection_keyframe(cur.ref(), pts.ref(), th, int(orb_dist), int(check_ori), a.ctypes.data)
    return n, a[:cur.view.N]


def oracle_fuse_candidates(kf, pts, th, inv_sigma2=None):
    L = oracle()
    L.orbo_fuse_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orbo_fuse_candidates.restype = None
    M = pts.view.M
    bi = np.full(max(M, 1), -1, np.int32); bd = np.full(max(M, 1), -1, np.int32)
    s2 = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
    L.orbo_fuse_candidates(kf.ref(), pts.ref(), th, int(s2 is not None), None if s2 is None else s2.ctypes.data, bi.ctypes.data, bd.ctypes.data)
    return bi[:M], bd[:M]


def oracle_search_by_sim3(kf1, kf2, p12, p21, th):
    L = oracle()
    L.orbo_search_by_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    m = np.full(max(kf1.view.N, 1), -1, np.int32)
    n = L.orbo_search_by_sim3(kf1.ref(), kf2.ref(), p12.ref(), p21.ref(), th, m.ctypes.data)
    return n, m[:kf1.view.N]


# ---- the reference's own DBoW2 (oracle/_ref/libref_dbow2.so) ----
_DBOW2 = None


def reference_dbow2():
    global _DBOW2
    if _DBOW2 is None:
        p = os.path.join(ORACLE_DIR, "_ref", "libref_dbow2.so")
        if not os.path.exists(p):
            build()
        if not os.path.exists(p):
            return None
        L = C.CDLL(p)
        L.ref_voc_load_text.restype = C.c_void_p; L.ref_voc_load_text.argtypes = [C.c_char_p]
        L.ref_voc_destroy.argtypes = [C.c_void_p]
        L.ref_voc_size.argtypes = [C.c_void_p]
        L.ref_voc_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 2 + [C.POINTER(C.c_int)] + [C.c_void_p] * 3 + [C.POINTER(C.c_int)]
        L.ref_voc_transform_one.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_double), C.POINTER(C.c_uint)]
        _DBOW2 = L
    return _DBOW2


class RefVocabulary:
    """ORBVocabulary of the reference, loaded with its own loadFromTextFile."""

    def __init__(self, path):
        self.L = reference_dbow2()
        self.h = self.L.ref_voc_load_text(str(path).encode())
        assert self.h, "reference loadFromTextFile failed"

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_voc_destroy(self.h)

    def size(self):
        return self.L.ref_voc_size(self.h)

    def transform(self, desc, levelsup=4):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); n = len(d); cap = max(n, 1)
        bi = np.zeros(cap, np.uint32); bv = np.zeros(cap, np.float64); fn = np.zeros(cap, np.uint32); fs = np.zeros(cap + 1, np.int32)
        ff = np.zeros(cap, np.uint32); nb, nf = C.c_int(), C.c_int()
        self.L.ref_voc_transform(self.h, d.ctypes.data, n, levelsup, bi.ctypes.data, bv.ctypes.data, C.byref(nb), fn.ctypes.data, fs.ctypes.data,
                                 ff.ctypes.data, C.byref(nf))
        return bi[:nb.value], bv[:nb.value], fn[:nf.value], fs[:nf.value + 1], ff[:int(fs[nf.value])]

    def transform_one(self, desc, levelsup=4):
        d = np.ascontiguousarray(desc, np.uint8); w = C.c_uint(); wt = C.c_double(); nd = C.c_uint()
        self.L.ref_voc_transform_one(self.h, d.ctypes.data, levelsup, C.byref(w), C.byref(wt), C.byref(nd))
        return w.value, wt.value, nd.value


def oracle_distinctive_descriptors(desc, start):
    L = oracle()
    L.orbo_distinctive_descriptors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.orbo_distinctive_descriptors.restype = None
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); st = np.ascontiguousarray(start, np.int32)
    best = np.full(max(len(st) - 1, 1), -1, np.int32)
    L.orbo_distinctive_descriptors(d.ctypes.data, st.ctypes.data, len(st) - 1, best.ctypes.data)
    return best[:len(st) - 1]


# ---- input pre-step primitives (cv::remap / cv::resize on cn channels / cv::cvtColor) ----
def oracle_remap(src, mapx, mapy):
    L = oracle()
    L.orbo_prim_remap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.orbo_prim_remap.restype = None
    src = np.ascontiguousarray(src, np.uint8); cn = 1 if src.ndim == 2 else src.shape[2]
    mx = np.ascontiguousarray(mapx, np.float32); my = np.ascontiguousarray(mapy, np.float32)
    dh, dw = mx.shape
    dst = np.zeros((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.uint8)
    L.orbo_prim_remap(src.ctypes.data, src.shape[1], src.shape[0], cn, mx.ctypes.data, my.ctypes.data, dst.ctypes.data, dw, dh)
    return dst


def oracle_resize_cn(src, dw, dh):
    L = oracle()
    L.orbo_prim_resize_cn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.orbo_prim_resize_cn.restype = None
    src = np.ascontiguousarray(src, np.uint8); cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.zeros((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.uint8)
    L.orbo_prim_resize_cn(src.ctypes.data, src.shape[1], src.shape[0], cn, dst.ctypes.data, dw, dh)
    return dst


def oracle_undistort(points, K, dist, variant=0):
    """cv::undistortPoints(points, K, dist, R = empty, P = K) for float32 [n, 2] points (oracle/orb_primitives.h)"""
    L = oracle()
    L.orbo_prim_undistort.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.orbo_prim_undistort.restype = None
    p = np.ascontiguousarray(points, np.float32).reshape(-1, 2); k = np.ascontiguousarray(K, np.float32); d = np.ascontiguousarray(dist, np.float32)
    out = np.zeros_like(p)
    L.orbo_prim_undistort(p.ctypes.data, len(p), k.ctypes.data, d.ctypes.data, len(d), int(variant), out.ctypes.data)
    return out


def oracle_gray(src, red_first, variant):
    L = oracle()
    L.orbo_prim_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orbo_prim_gray.restype = None
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros(src.shape[:2], np.uint8)
    L.orbo_prim_gray(src.ctypes.data, src.shape[1], src.shape[0], src.shape[2], int(red_first), int(variant), dst.ctypes.data)
    return dst


def oracle_search_by_projection_mappoints_fisheye(frame2, mps, mps_r, th, bFar, thFar, nnratio):
    L = oracle()
    L.orbo_search_by_projection_mappoints_fisheye.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p]
    N = frame2.view.left.N + frame2.view.right.N
    a = np.full(max(N, 1), -1, np.int32)
    n = L.orbo_search_by_projection_mappoints_fisheye(frame2.ref(), mps.ref(), mps_r.ref(), th, int(bFar), thFar, nnratio, a.ctypes.data)
    return n, a[:N]


def oracle_search_by_projection_frame_fisheye(cur2, last, proj_ur, proj_vr, th, fwd, bwd, check_ori):
    L = oracle()
    L.orbo_search_by_projection_frame_fisheye.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]
    N = cur2.view.left.N + cur2.view.right.N
    ur = np.ascontiguousarray(proj_ur, np.float32); vr = np.ascontiguousarray(proj_vr, np.float32)
    a = np.full(max(N, 1), -1, np.int32)
    n = L.orbo_search_by_projection_frame_fisheye(cur2.ref(), last.ref(), ur.ctypes.data, vr.ctypes.data, th, int(fwd), int(bwd), int(check_ori), a.ctypes.data)
    return n, a[:N]


_REF_FRAME = None


def _bind_frame_lib(L):
        L.ref_frame_stereo.restype = C.c_void_p
        L.ref_frame_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_float] * 6 + [C.POINTER(C.c_int)] * 2
        L.ref_frame_destroy.argtypes = [C.c_void_p]
        L.ref_frame_rgbd.restype = C.c_void_p
        L.ref_frame_rgbd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_float] * 6 + [C.c_void_p, C.POINTER(C.c_int)]
        L.ref_frame_fisheye.restype = C.c_void_p
        L.ref_frame_fisheye.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p]
        L.ref_frame_fisheye_get3d.argtypes = [C.c_void_p] * 3
        L.ref_frame_fisheye_get.argtypes = [C.c_void_p] * 6
        L.ref_frame_stereo_repeat.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_float] * 6 + [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
        L.ref_frame_get.argtypes = [C.c_void_p] * 8
        L.ref_frame_constants.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_frame_features_in_area.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]
        return L


def reference_frame_lib():
    """The reference's own Frame.cc + ORBextractor.cc + ORBmatcher.cc built over oracle/slam_shim/frame_world.h (oracle/_ref/libref_frame.so); None if absent."""
    global _REF_FRAME
    if _REF_FRAME is None:
        p = os.path.join(ORACLE_DIR, "_ref", "libref_frame.so")
        if not os.path.exists(p) and os.path.exists("/root/reference/src/Frame.cc"):
            build()
        if not os.path.exists(p):
            return None
        _REF_FRAME = _bind_frame_lib(C.CDLL(p))
    return _REF_FRAME


_REF_FRAME_KNN = None


def reference_frame_knn_lib():
    """oracle/_ref/libref_frame_knn.so: the Frame driver built with a Kannala-Brandt camera whose TriangulateMatches accepts every pair (the
    reference's KannalaBrandt8.cpp left out; oracle/ref_frame_driver.cpp, ORBX_KB8_ACCEPT_ALL)."""
    global _REF_FRAME_KNN
    if _REF_FRAME_KNN is None:
        p = os.path.join(ORACLE_DIR, "_ref", "libref_frame_knn.so")
        if not os.path.exists(p) and os.path.exists("/root/reference/src/Frame.cc"):
            build()
        _REF_FRAME_KNN = _bind_frame_lib(C.CDLL(p))
    return _REF_FRAME_KNN


def dropin_frame_lib(orbx_path):
    """oracle/_ref/libref_frame_dropin.so: the reference's own Frame.cc compiled against the drop-in ORBextractor.h, i.e. the
    reference's stereo Frame constructor running on the product library `orbx_path` (HIP or emulator build).  Load once per process."""
    C.CDLL(orbx_path, mode=C.RTLD_GLOBAL)
    return _bind_frame_lib(C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libref_frame_dropin.so")))


class ReferenceFrame:
    """ORB_SLAM3::Frame as built by the reference's own stereo constructor (src/Frame.cc:105-230) on a rectified pair."""

    def __init__(self, left, right, nfeatures=1200, scale=1.2, nlevels=8, ini=20, mn=7, gauss_variant=0, fx=458.654, fy=457.296, cx=367.215, cy=248.375, bf=458.654 * 0.110074, th_depth=35.0,
                 lib=None, depth=None, dist=None):
        """depth given (float32 [H, W]): the RGB-D constructor (src/Frame.cc:235-345) on (left = grey image, depth); `right` is ignored;
        dist = (k1, k2, p1, p2, k3): lens distortion (UndistortKeyPoints / ComputeImageBounds through the restated cv::undistortPoints)"""
        L = lib or reference_frame_lib()
        left = np.ascontiguousarray(left, np.uint8)
        n = C.c_int(); nr = C.c_int()
        self.L = L
        if depth is not None:
            depth = np.ascontiguousarray(depth, np.float32)
            d5 = None if dist is None else np.ascontiguousarray(dist, np.float32)
            assert d5 is None or d5.size == 5
            self.h = L.ref_frame_rgbd(left.ctypes.data, depth.ctypes.data, left.shape[1], left.shape[0], nfeatures, scale, nlevels, ini, mn, gauss_variant,
                                      fx, fy, cx, cy, bf, th_depth, None if d5 is None else d5.ctypes.data, C.byref(n))
        else:
            right = np.ascontiguousarray(right, np.uint8)
            self.h = L.ref_frame_stereo(left.ctypes.data, right.ctypes.data, left.shape[1], left.shape[0], nfeatures, scale, nlevels, ini, mn, gauss_variant,
                                        fx, fy, cx, cy, bf, th_depth, C.byref(n), C.byref(nr))
        N, NR = n.value, nr.value
        self.N = N
        self.keys = np.zeros(N, KP_DTYPE); self.keys_un = np.zeros(N, KP_DTYPE); self.desc = np.zeros((N, 32), np.uint8)
        self.u_right = np.zeros(N, np.float32); self.depth = np.zeros(N, np.float32)
        self.keys_right = np.zeros(NR, KP_DTYPE); self.desc_right = np.zeros((NR, 32), np.uint8)
        L.ref_frame_get(self.h, self.keys.ctypes.data, self.keys_un.ctypes.data, self.desc.ctypes.data, self.u_right.ctypes.data, self.depth.ctypes.data,
                        self.keys_right.ctypes.data, self.desc_right.ctypes.data)
        c = np.zeros(8, np.float32); L.ref_frame_constants(self.h, c.ctypes.data)
        self.bounds = c[:4].copy(); self.grid_inv = c[4:6].copy(); self.mbf, self.mb = float(c[6]), float(c[7])

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_frame_destroy(self.h); self.h = None

    def search_local_points(self, R, t, pos, normal, min_dist, max_dist, bad, has_obs, desc, cos_limit=0.5, search=True, th=1.0, far_points=False, th_far=50.0, nnratio=0.8):
        """Frame::SetPose + Frame::isInFrustum for every point + ORBmatcher::SearchByProjection(F, points, th, bFarPoints, thFarPoints) on the
        reference's own code.  Returns (track dict, assigned[N], nmatches)."""
        M = len(pos)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        R, t, pos, normal, min_dist, max_dist = f32(R), f32(t), f32(pos), f32(normal), f32(min_dist), f32(max_dist)
        bad = np.ascontiguousarray(bad, np.uint8); has_obs = np.ascontiguousarray(has_obs, np.uint8); desc = np.ascontiguousarray(desc, np.uint8)
        track = np.zeros((7, max(M, 1)), np.float32); assigned = np.full(max(self.N, 1), -1, np.int32)
        self.L.ref_frame_search_local_points.restype = C.c_int
        self.L.ref_frame_search_local_points.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 7 + [C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p]
        n = self.L.ref_frame_search_local_points(self.h, R.ctypes.data, t.ctypes.data, M, pos.ctypes.data, normal.ctypes.data, min_dist.ctypes.data, max_dist.ctypes.data,
                                                 bad.ctypes.data, has_obs.ctypes.data, desc.ctypes.data, cos_limit, track.ctypes.data, int(search), th, int(far_points), th_far, nnratio,
                                                 assigned.ctypes.data)
        tr = dict(in_view=track[0, :M] > 0, proj_x=track[1, :M], proj_y=track[2, :M], proj_xr=track[3, :M], depth=track[4, :M], view_cos=track[5, :M],
                  scale_level=track[6, :M].astype(np.int32))
        return tr, assigned[:self.N], n

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        idx = np.zeros(max(self.N, 1), np.int32)
        n = self.L.ref_frame_features_in_area(self.h, x, y, r, min_level, max_level, idx.ctypes.data, len(idx))
        return idx[:n].copy()

    def search_keyframe(self, R, t, pos, kind, min_dist, max_dist, angle, desc, th, orb_dist, check_orientation=True, nnratio=0.9, occupied=None):
        """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (relocalisation) on the reference's own Frame + ORBmatcher.cc.
        kind[i]: 0 no map point, 1 good, 2 bad, 3 in sAlreadyFound.  Returns (nmatches, assigned[N])."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        u8 = lambda a: np.ascontiguousarray(a, np.uint8)
        R, t, pos, min_dist, max_dist, angle = f32(R), f32(t), f32(pos), f32(min_dist), f32(max_dist), f32(angle)
        kind, desc = u8(kind), u8(desc)
        occ = None if occupied is None else u8(occupied)
        assigned = np.full(max(self.N, 1), -1, np.int32)
        fn = self.L.ref_frame_search_keyframe
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        n = fn(self.h, R.ctypes.data, t.ctypes.data, len(pos), pos.ctypes.data, kind.ctypes.data, min_dist.ctypes.data, max_dist.ctypes.data, angle.ctypes.data, desc.ctypes.data,
               float(th), int(orb_dist), int(check_orientation), float(nnratio), None if occ is None else occ.ctypes.data, assigned.ctypes.data)
        return n, assigned[:self.N]

    def search_lastframe(self, R, t, Rl, tl, pos, valid, octave, angle, has_obs, desc, th, mono, check_orientation=True, nnratio=0.9, occupied=None):
        """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) on the reference's own Frame + ORBmatcher.cc: this frame at pose (R, t) against
        a last frame at (Rl, tl) with the given map points.  Returns (nmatches, assigned[N], bForward, bBackward)."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        u8 = lambda a: np.ascontiguousarray(a, np.uint8)
        R, t, Rl, tl, pos, angle = f32(R), f32(t), f32(Rl), f32(tl), f32(pos), f32(angle)
        valid, has_obs, desc = u8(valid), u8(has_obs), u8(desc)
        octave = np.ascontiguousarray(octave, np.int32)
        occ = None if occupied is None else u8(occupied)
        assigned = np.full(max(self.N, 1), -1, np.int32); fb = np.zeros(2, np.int32)
        fn = self.L.ref_frame_search_lastframe
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        n = fn(self.h, R.ctypes.data, t.ctypes.data, Rl.ctypes.data, tl.ctypes.data, len(pos), pos.ctypes.data, valid.ctypes.data, octave.ctypes.data, angle.ctypes.data,
               has_obs.ctypes.data, desc.ctypes.data, float(th), int(mono), int(check_orientation), float(nnratio), None if occ is None else occ.ctypes.data,
               assigned.ctypes.data, fb.ctypes.data)
        return n, assigned[:self.N], bool(fb[0]), bool(fb[1])


def reference_frame_repeat(left, right, seconds, nfeatures=1200, scale=1.2, nlevels=8, ini=20, mn=7, fx=458.654, fy=457.296, cx=367.215, cy=248.375, bf=458.654 * 0.110074, th_depth=35.0):
    """Constructs the reference's stereo Frame on (left, right) over and over for `seconds` (long-lived extractors, as Tracking holds them).
    Returns (frames, elapsed seconds, stereo matches of the last frame, sum of "ORB Extraction" ms, sum of "Stereo Matching" ms).  Releases the GIL."""
    L = reference_frame_lib()
    left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
    el = C.c_double(); m = C.c_int()
    st = np.zeros(2, np.float64)          # sums of the reference's own timers: mTimeORB_Ext, mTimeStereoMatch (ms; REGISTER_TIMES build)
    n = L.ref_frame_stereo_repeat(left.ctypes.data, right.ctypes.data, left.shape[1], left.shape[0], nfeatures, scale, nlevels, ini, mn, fx, fy, cx, cy, bf, th_depth,
                                  float(seconds), C.byref(el), C.byref(m), st.ctypes.data)
    return n, el.value, m.value, float(st[0]), float(st[1])


_REF_MP = None


def reference_mappoint_lib():
    """The reference's own MapPoint.cc (+ Frame.cc, ORBmatcher.cc) built over oracle/slam_shim/mappoint_world.h (oracle/_ref/libref_mappoint.so); None if absent."""
    global _REF_MP
    if _REF_MP is None:
        p = os.path.join(ORACLE_DIR, "_ref", "libref_mappoint.so")
        if not os.path.exists(p) and os.path.exists("/root/reference/src/MapPoint.cc"):
            build()
        if not os.path.exists(p):
            return None
        L = C.CDLL(p)
        L.ref_mp_distinctive.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _REF_MP = L
    return _REF_MP


def reference_distinctive_descriptors(desc, start, right_of_prev=None, bad_kf=None):
    """MapPoint::ComputeDistinctiveDescriptors as the reference's own MapPoint.cc executes it.  Returns (descriptors [P,32], has [P])."""
    L = reference_mappoint_lib()
    desc = np.ascontiguousarray(desc, np.uint8); start = np.ascontiguousarray(start, np.int32)
    n = len(desc); P = len(start) - 1
    rp = np.zeros(max(n, 1), np.uint8) if right_of_prev is None else np.ascontiguousarray(right_of_prev, np.uint8)
    bk = np.zeros(max(n, 1), np.uint8) if bad_kf is None else np.ascontiguousarray(bad_kf, np.uint8)
    out = np.zeros((max(P, 1), 32), np.uint8); has = np.zeros(max(P, 1), np.uint8)
    d = desc if n else np.zeros((1, 32), np.uint8)
    L.ref_mp_distinctive(d.ctypes.data, start.ctypes.data, rp.ctypes.data, bk.ctypes.data, P, out.ctypes.data, has.ctypes.data)
    return out[:P], has[:P]


class ReferenceRigFrame:
    """The reference's fisheye-rig Frame (src/Frame.cc:1432-1528) kept alive for Frame::isInFrustum (two cameras) and
    ORBmatcher::SearchByProjection(F, points, ...) with its right-camera branch.  cams = (cam1[8], cam2[8], Rlr[3,3], tlr[3])."""

    def __init__(self, left, right, lap_left, lap_right, nfeatures, cams, scale=1.2, nlevels=8, ini=20, mn=7, lib=None):
        L = self.L = lib or reference_frame_lib()
        left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
        out = np.zeros(4, np.int32)
        cp = np.concatenate([np.asarray(c, np.float32).ravel() for c in cams]).astype(np.float32); assert cp.size == 28
        self.h = L.ref_frame_fisheye(left.ctypes.data, right.ctypes.data, left.shape[1], left.shape[0], nfeatures, scale, nlevels, ini, mn, 0,
                                     lap_left[0], lap_left[1], lap_right[0], lap_right[1], cp.ctypes.data, out.ctypes.data)
        self.nl, self.nr, self.mono_left, self.mono_right = [int(v) for v in out]
        self.keys = np.zeros(self.nl, KP_DTYPE); self.keys_right = np.zeros(self.nr, KP_DTYPE); self.desc = np.zeros((self.nl + self.nr, 32), np.uint8)
        self.l2r = np.zeros(max(self.nl, 1), np.int32); self.r2l = np.zeros(max(self.nr, 1), np.int32)
        L.ref_frame_fisheye_get(self.h, self.keys.ctypes.data, self.keys_right.ctypes.data, self.desc.ctypes.data, self.l2r.ctypes.data, self.r2l.ctypes.data)
        self.l2r = self.l2r[:self.nl]; self.r2l = self.r2l[:self.nr]

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_frame_destroy(self.h); self.h = None

    def search_local_points(self, R, t, pos, normal, min_dist, max_dist, bad, has_obs, desc, cos_limit=0.5, search=True, th=1.0, far_points=False, th_far=50.0, nnratio=0.8):
        """Returns (left dict, right dict, assigned [Nleft + Nright], nmatches, pose dict as the Frame holds it after SetPose)."""
        M = len(pos)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        R, t, pos, normal, min_dist, max_dist = f32(R), f32(t), f32(pos), f32(normal), f32(min_dist), f32(max_dist)
        bad = np.ascontiguousarray(bad, np.uint8); has_obs = np.ascontiguousarray(has_obs, np.uint8); desc = np.ascontiguousarray(desc, np.uint8)
        track = np.zeros((13, max(M, 1)), np.float32); assigned = np.full(max(self.nl + self.nr, 1), -1, np.int32); pose = np.zeros(45, np.float32)
        fn = self.L.ref_frame_search_local_points_rig
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 7 + [C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        n = fn(self.h, R.ctypes.data, t.ctypes.data, M, pos.ctypes.data, normal.ctypes.data, min_dist.ctypes.data, max_dist.ctypes.data, bad.ctypes.data, has_obs.ctypes.data,
               desc.ctypes.data, cos_limit, track.ctypes.data, int(search), th, int(far_points), th_far, nnratio, assigned.ctypes.data, pose.ctypes.data)
        left = dict(in_view=track[0, :M] > 0, proj_x=track[1, :M], proj_y=track[2, :M], depth=track[3, :M], view_cos=track[4, :M], scale_level=track[5, :M].astype(np.int32))
        right = dict(in_view_r=track[6, :M] > 0, proj_xr=track[7, :M], proj_yr=track[8, :M], depth_r=track[9, :M], view_cos_r=track[10, :M], scale_level_r=track[11, :M].astype(np.int32))
        p = dict(Rcw=pose[0:9].reshape(3, 3), tcw=pose[9:12], Ow=pose[12:15], Rwc=pose[15:24].reshape(3, 3), Rrl=pose[24:33].reshape(3, 3), trl=pose[33:36], tlr=pose[36:39])
        return left, right, assigned[:self.nl + self.nr], n, p


def reference_frame_fma_lib():
    """oracle/_ref/libref_frame_fma.so: the frame driver built as the reference's CMakeLists.txt builds on an FMA host (-O3, AVX2 + FMA, contraction on); None if absent"""
    p = os.path.join(ORACLE_DIR, "_ref", "libref_frame_fma.so")
    return _bind_frame_lib(C.CDLL(p)) if os.path.exists(p) else None


def reference_fisheye_frame(left, right, lap_left, lap_right, nfeatures=1500, scale=1.2, nlevels=8, ini=20, mn=7, gauss_variant=0, cams=None, lib=None):
    """The reference's fisheye-rig Frame constructor (src/Frame.cc:1432-1528).  cams = (cam1[8], cam2[8], Rlr[3,3], tlr[3]): the gate is the
    reference's own KannalaBrandt8::TriangulateMatches (src/CameraModels/KannalaBrandt8.cpp compiled into libref_frame.so); cams = None: the
    accept-all build (libref_frame_knn.so), whose result is the kNN + ratio decision alone.
    Returns dict(keys, keys_right, desc [Nleft+Nright,32], mono_left, mono_right, l2r, r2l, depth, p3d)."""
    L = lib or (reference_frame_lib() if cams is not None else reference_frame_knn_lib())
    left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
    out = np.zeros(4, np.int32)
    cp = None
    if cams is not None:
        cp = np.concatenate([np.asarray(c, np.float32).ravel() for c in cams]).astype(np.float32); assert cp.size == 28
    h = L.ref_frame_fisheye(left.ctypes.data, right.ctypes.data, left.shape[1], left.shape[0], nfeatures, scale, nlevels, ini, mn, gauss_variant,
                            lap_left[0], lap_left[1], lap_right[0], lap_right[1], None if cp is None else cp.ctypes.data, out.ctypes.data)
    assert h, "ref_frame_fisheye refused (camera mode of the library does not match)"
    nl, nr, ml, mr = [int(v) for v in out]
    keys = np.zeros(nl, KP_DTYPE); keys_r = np.zeros(nr, KP_DTYPE); desc = np.zeros((nl + nr, 32), np.uint8)
    l2r = np.zeros(max(nl, 1), np.int32); r2l = np.zeros(max(nr, 1), np.int32)
    L.ref_frame_fisheye_get(h, keys.ctypes.data, keys_r.ctypes.data, desc.ctypes.data, l2r.ctypes.data, r2l.ctypes.data)
    depth = np.zeros(max(nl, 1), np.float32); p3d = np.zeros((max(nl, 1), 3), np.float32)
    L.ref_frame_fisheye_get3d(h, depth.ctypes.data, p3d.ctypes.data)
    L.ref_frame_destroy(h)
    return dict(keys=keys, keys_right=keys_r, desc=desc, mono_left=ml, mono_right=mr, l2r=l2r[:nl], r2l=r2l[:nr], depth=depth[:nl], p3d=p3d[:nl])
