"""Host-side mirror of ``ORBVocabulary`` (``DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>``, include/ORBVocabulary.h:28-29)
for the one thing the front-end asks of it: ``transform(features, BowVector&, FeatureVector&, levelsup)``
(Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1195), called by ``Frame::ComputeBoW`` (src/Frame.cc:984-997)."""
import ctypes as C

import numpy as np

# DBoW2::ScoringType / WeightingType, Thirdparty/DBoW2/DBoW2/BowVector.h:37-56
L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = range(6)
TF_IDF, TF, IDF, BINARY = range(4)


class BowResult:
    """mBowVec as (ids, values), mFeatVec as CSR (nodes, start, features), plus the per-feature words / nodes."""

    def __init__(self, bow_id, bow_val, fv_node, fv_start, fv_feat, word_id, node_id):
        self.bow_id, self.bow_val, self.fv_node, self.fv_start, self.fv_feat = bow_id, bow_val, fv_node, fv_start, fv_feat
        self.word_id, self.node_id = word_id, node_id


class ORBVocabulary:
    def __init__(self, ext, handle):
        self._ext, self._lib, self._v = ext, ext._lib, handle

    @classmethod
    def from_arrays(cls, ext, k, L, scoring, weighting, parent, is_leaf, desc, weight):
        """Nodes in ORBvoc.txt line order: entry i is node i + 1 (parent 0 = root)."""
        p = np.ascontiguousarray(parent, np.int32); lf = np.ascontiguousarray(is_leaf, np.uint8)
        d = np.ascontiguousarray(desc, np.uint8).reshape(len(p), 32); w = np.ascontiguousarray(weight, np.float64)
        h = C.c_void_p()
        ext._lib.check(ext._lib.L.orbv_create(ext._h, int(k), int(L), int(scoring), int(weighting), len(p), p.ctypes.data, lf.ctypes.data,
                                              d.ctypes.data, w.ctypes.data, C.byref(h)))
        return cls(ext, h)

    @classmethod
    def loadFromTextFile(cls, ext, path):
        h = C.c_void_p()
        ext._lib.check(ext._lib.L.orbv_load_text(ext._h, str(path).encode(), C.byref(h)))
        return cls(ext, h)

    def close(self):
        if self._v:
            self._lib.L.orbv_destroy(self._v)
            self._v = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return self._lib.L.orbv_words(self._v)

    def _alloc(self, cap):
        return (np.zeros(cap, np.uint32), np.zeros(cap, np.float64), np.zeros(cap, np.uint32), np.zeros(cap + 1, np.int32),
                np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32))

    @staticmethod
    def _pack(arrs, n, nb, nf):
        bi, bv, fn, fs, ff, wi, ni = arrs
        return BowResult(bi[:nb].copy(), bv[:nb].copy(), fn[:nf].copy(), fs[:nf + 1].copy(), ff[:int(fs[nf])].copy(), wi[:n].copy(), ni[:n].copy())

    def transform(self, desc, levelsup=4):
        """transform(features, BowVector&, FeatureVector&, levelsup) on n host descriptors [n, 32]."""
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        a = self._alloc(max(n, 1))
        nb, nf = C.c_int(), C.c_int()
        self._lib.check(self._lib.L.orbv_transform(self._v, self._ext._h, d.ctypes.data, n, int(levelsup), a[5].ctypes.data, a[6].ctypes.data,
                                                   a[0].ctypes.data, a[1].ctypes.data, C.byref(nb), a[2].ctypes.data, a[3].ctypes.data,
                                                   a[4].ctypes.data, C.byref(nf)))
        return self._pack(a, n, nb.value, nf.value)

    def transform_extracted(self, ext, first=0, B=None, levelsup=4):
        """The same for images [first, first+B) of ext's last batch, on the device-resident descriptors (asynchronous)."""
        B = ext._B - first if B is None else B
        self._lib.check(self._lib.L.orbv_transform_extracted(self._v, ext._h, int(first), int(B), int(levelsup)))

    def fetch(self, ext, b, n_features):
        a = self._alloc(max(ext.max_keypoints(), 1))
        nb, nf = C.c_int(), C.c_int()
        self._lib.check(self._lib.L.orbv_fetch(self._v, ext._h, int(b), a[5].ctypes.data, a[6].ctypes.data, int(n_features), a[0].ctypes.data,
                                               a[1].ctypes.data, C.byref(nb), a[2].ctypes.data, a[3].ctypes.data, a[4].ctypes.data, C.byref(nf)))
        return self._pack(a, n_features, nb.value, nf.value)
