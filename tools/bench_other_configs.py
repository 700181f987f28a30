"""Throughput of the other BASELINE.json configurations (they are parity cases, not the headline metric): one handle, 128 frames per
batch resident in HBM, results fetched to the host, best of a few batches.  Output: JSON on stdout."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from orb_slam3_detailed_comments_amd import synth, ORBextractor
from orb_slam3_detailed_comments_amd import matcher as M


def time_batches(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


class Direct:
    """fetch straight into pinned buffers through the C ABI (what bench.py does); the convenience methods of the Python mirror allocate
    and repack per call and would dominate these timings"""

    def __init__(self, ex, B):
        self.ex, self.B, self.cap = ex, B, ex.max_keypoints()
        self.k = ex.pinned_empty((B, self.cap, 28), np.uint8); self.d = ex.pinned_empty((B, self.cap, 32), np.uint8)
        self.n = np.zeros(B, np.int32); self.m = np.zeros(B, np.int32)

    def fetch(self):
        L = self.ex._lib
        L.check(L.L.orbx_fetch(self.ex._h, self.k.ctypes.data, self.d.ctypes.data, self.cap, self.n.ctypes.data, self.m.ctypes.data))


def main():
    out = {}
    B = 128
    # configs[0]: monocular EuRoC, nFeatures 1000, lapping {0, 1000}
    imgs = np.stack([synth.corner_field(752, 480, seed=i % 16) for i in range(B)])
    ex = ORBextractor(1000, 1.2, 8, 20, 7); dp = ex.device_upload(imgs); f1 = Direct(ex, B)
    def mono():
        ex.enqueue(None, (0, 1000), device_ptr=dp, shape=imgs.shape); f1.fetch()
    out["mono 752x480 n1000 lap{0,1000}"] = {"frames_per_s": round(B / time_batches(mono), 1)}
    # as shipped: 752x480 -> 600x350 (cv::resize pre-step on the device), nFeatures 1000
    ex2 = ORBextractor(1000, 1.2, 8, 20, 7); ex2.set_input(1, resize=(600, 350)); dp2 = ex2.device_upload(imgs)
    ex2.enqueue(None, (0, 1000), device_ptr=dp2, shape=imgs.shape); f2 = Direct(ex2, B)
    def mono_small():
        ex2.enqueue(None, (0, 1000), device_ptr=dp2, shape=imgs.shape); f2.fetch()
    out["mono EuRoC as shipped: resize 752x480 -> 600x350, n1000"] = {"frames_per_s": round(B / time_batches(mono_small), 1)}
    # monocular initialisation extractor: 5 * nFeatures
    ex3 = ORBextractor(5000, 1.2, 8, 20, 7); dp3 = ex3.device_upload(imgs[:32]); f3 = Direct(ex3, 32)
    def mono_init():
        ex3.enqueue(None, (0, 1000), device_ptr=dp3, shape=imgs[:32].shape); f3.fetch()
    out["mono init 752x480 n5000"] = {"frames_per_s": round(32 / time_batches(mono_init), 1)}
    # configs[2]: TUM-VI fisheye stereo 512x512, nFeatures 1500, lapping {0, 511}: extract L+R + BFMatcher kNN(2) + ratio
    pairs = [synth.stereo_pair(512, 512, seed=i % 16) for i in range(B // 2)]
    batch = np.stack([p[0] for p in pairs] + [p[1] for p in pairs])
    ex4 = ORBextractor(1500, 1.2, 8, 20, 7); dp4 = ex4.device_upload(batch); f4 = Direct(ex4, B)
    kn = [ex4.pinned_empty((B // 2, f4.cap), np.int32) for _ in range(4)]; kr = ex4.pinned_empty((B // 2, f4.cap), np.uint8)
    def fisheye():
        L = ex4._lib
        ex4.enqueue(None, (0, 511), device_ptr=dp4, shape=batch.shape)
        L.check(L.L.orbm_knn2(ex4._h, 0, ex4._h, B // 2, B // 2))
        f4.fetch()
        L.check(L.L.orbm_knn2_fetch(ex4._h, B // 2, kn[0].ctypes.data, kn[1].ctypes.data, kn[2].ctypes.data, kn[3].ctypes.data, kr.ctypes.data, f4.cap))
    out["TUM-VI 512x512 stereo n1500 lap{0,511}: extract L+R + 2-NN + ratio"] = {"pairs_per_s": round((B // 2) / time_batches(fisheye), 1)}
    # configs[3]: TUM RGB-D 640x480 RGB frames, nFeatures 1000 (cvtColor on the device)
    rgb = np.stack([np.stack([synth.corner_field(640, 480, seed=(i + 7 * c) % 16, nrect=2550) for c in range(3)], axis=2) for i in range(B)])
    ex5 = ORBextractor(1000, 1.2, 8, 20, 7); ex5.set_input(3, rgb=True); dp5 = ex5.device_upload(rgb); f5 = Direct(ex5, B)
    def rgbd():
        ex5.enqueue(None, (0, 0), device_ptr=dp5, shape=rgb.shape[:3], stride=640 * 3); f5.fetch()
    out["TUM RGB-D 640x480 RGB -> grey on the device, n1000"] = {"frames_per_s": round(B / time_batches(rgbd), 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
