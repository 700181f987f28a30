"""Does v_pk_maximum3_f16 / v_pk_minimum3_f16 order positive binary16 DENORMAL patterns (0x0000..0x00FF in each half) like integers?
(kernel FP mode: fp16 denormals preserved).  Uses the instruction self-test entry."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
N = 1 << 16
rng = np.random.default_rng(5)
pix = lambda: (rng.integers(0, 256, N) | rng.integers(0, 256, N) << 16).astype(np.uint32)
a, b, c = pix(), pix(), pix()
a[:8] = [0, 0, 0x00FF00FF, 0x00010000, 0x00000001, 0x00FF0000, 0x000000FF, 0x00800080]
ex = ORBextractor(500, 1.2, 8, 20, 7)
out = np.zeros((20, N), np.uint32)
ex._lib.check(ex._lib.L.orbx_debug_simd_selftest(ex._h, a.ctypes.data, b.ctypes.data, c.ctypes.data, N, out.ctypes.data))
h = lambda x: ((x & 0xFFFF).astype(np.int64), (x >> 16).astype(np.int64))
hs = [h(x) for x in (a, b, c)]
mx = np.maximum.reduce([t[0] for t in hs]) | np.maximum.reduce([t[1] for t in hs]) << 16
mn = np.minimum.reduce([t[0] for t in hs]) | np.minimum.reduce([t[1] for t in hs]) << 16
print("pk_max3 on denormal patterns: %d mismatches of %d" % (int((out[6] != mx.astype(np.uint32)).sum()), N))
print("pk_min3 on denormal patterns: %d mismatches of %d" % (int((out[7] != mn.astype(np.uint32)).sum()), N))
bad = np.nonzero(out[6] != mx.astype(np.uint32))[0][:5]
for i in bad: print(hex(a[i]), hex(b[i]), hex(c[i]), "->", hex(out[6][i]), "expected", hex(int(mx[i])))
