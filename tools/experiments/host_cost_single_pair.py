import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam3_detailed_comments_amd import synth, load_hip
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
lib = load_hip()
L, R = synth.stereo_pair(seed=100)
pair = np.stack([L, R])
bf, b = 458.654 * 0.110074, 0.110074
ex = ORBextractor(1200, 1.2, 8, 20, 7)
ptr, shp, strd, istrd = ex.input_upload(pair)
for it in range(20):
    ex.enqueue(None, (0, 0), device_ptr=ptr, shape=shp, stride=strd, image_stride=istrd); lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 1, 1, bf, b)); ex.sync()
N = 300
te = tm = ts = 0.0
for it in range(N):
    t0 = time.perf_counter()
    ex.enqueue(None, (0, 0), device_ptr=ptr, shape=shp, stride=strd, image_stride=istrd)
    t1 = time.perf_counter()
    lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 1, 1, bf, b))
    t2 = time.perf_counter()
    ex.sync()
    t3 = time.perf_counter()
    te += t1 - t0; tm += t2 - t1; ts += t3 - t2
print("host time per pair: enqueue (5 launches) %.1f us, stereo_match (2 launches) %.1f us, sync wait %.1f us, total %.1f us" % (te / N * 1e6, tm / N * 1e6, ts / N * 1e6, (te + tm + ts) / N * 1e6))
