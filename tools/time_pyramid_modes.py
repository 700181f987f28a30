"""Pyramid stage time (serial stage profile) and whole-batch latency for both launch forms of ComputePyramid over a range of batch sizes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam3_detailed_comments_amd import synth
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
from orb_slam3_detailed_comments_amd import _lib
LIB = _lib.OrbxLib(os.environ["ORBX_BENCH_LIB"]) if os.environ.get("ORBX_BENCH_LIB") else None      # a variant build (tools/experiments/sweep_tunables.py)
BATCHES = [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16, 32, 64]
imgs = np.stack([synth.stereo_pair(seed=100 + s)[s & 1] for s in range(64)])
for B in BATCHES:
    for mode in (1, 2):
        ex = ORBextractor(1200, 1.2, 8, 20, 7, lib=LIB)
        ex.pyramid_mode(mode)
        arr = imgs[:B]
        dptr = ex.device_upload(arr)
        ex.profile(True, serial=True)
        for it in range(3):
            ex.enqueue(None, (0, 0), device_ptr=dptr, shape=arr.shape); ex.sync()
        acc = 0.0
        K = 20
        for it in range(K):
            ex.enqueue(None, (0, 0), device_ptr=dptr, shape=arr.shape); ex.sync()
            acc += ex.stage_ms()["pyramid"] / K
        ex.profile(False)
        t = time.perf_counter()
        for it in range(200):
            ex.enqueue(None, (0, 0), device_ptr=dptr, shape=arr.shape); ex.sync()
        lat = (time.perf_counter() - t) / 200 * 1e3
        print("B=%3d mode %d (%s): pyramid stage %.4f ms, extraction of the batch %.4f ms" % (B, mode, "per level" if mode == 1 else "one launch", acc, lat), flush=True)
        ex.close()
