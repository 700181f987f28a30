"""One stereo pair per call, synchronised after every pair (Tracking's rhythm), frames written into pyramid level 0 by the producer:
the loop that tests/gpu_quick.py times, alone, for a rocprofv3 kernel trace of the launch chain."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam3_detailed_comments_amd import synth, load_hip
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
lib = load_hip()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
L, R = synth.stereo_pair(seed=100)
pair = np.stack([L, R])
bf, b = 458.654 * 0.110074, 0.110074
ex = ORBextractor(1200, 1.2, 8, 20, 7)
ptr, shp, strd, istrd = ex.input_upload(pair)
for it in range(10):
    ex.enqueue(None, (0, 0), device_ptr=ptr, shape=shp, stride=strd, image_stride=istrd); lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 1, 1, bf, b)); ex.sync()
t = time.time()
for it in range(N):
    ex.enqueue(None, (0, 0), device_ptr=ptr, shape=shp, stride=strd, image_stride=istrd); lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 1, 1, bf, b)); ex.sync()
print("single pair, zero-copy input: %.3f ms per pair over %d pairs" % ((time.time() - t) / N * 1e3, N))
