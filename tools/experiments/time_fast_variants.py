"""Times k_fast_cells (serial stage profile, 128 images of the bench workload) for every build/variants/liborbx_hip_*.so."""
import glob, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam3_detailed_comments_amd import _lib, synth
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
B = 128
imgs = []
for s in range(B // 2):
    l, r = synth.stereo_pair(seed=100 + s); imgs += [l]
for s in range(B // 2):
    l, r = synth.stereo_pair(seed=100 + s); imgs += [r]
arr_big = np.stack(imgs)
arr_small = np.stack([imgs[0], imgs[B // 2]])
paths = sorted(glob.glob(os.path.join(ROOT, "build", "variants", "liborbx_hip_*.so")))
if len(sys.argv) > 1: paths = [p for p in paths if any(a in p for a in sys.argv[1:])]
for path, arr in [(p, a) for p in paths for a in (arr_big, arr_small)]:
    lib = _lib.OrbxLib(path)
    ex = ORBextractor(1200, 1.2, 8, 20, 7, lib=lib)
    dptr = ex.device_upload(arr)
    ex.profile(True, serial=True)
    for it in range(3):
        ex.enqueue(None, (0, 0), device_ptr=dptr, shape=arr.shape); ex.sync()
    acc = {}
    K = 10
    for it in range(K):
        ex.enqueue(None, (0, 0), device_ptr=dptr, shape=arr.shape); ex.sync()
        for k, v in ex.stage_ms().items(): acc[k] = acc.get(k, 0.0) + v / K
    print("%-28s B=%3d fast_cells %.4f ms   (pyramid %.4f quadtree %.3f blur %.3f orient %.3f)" % (os.path.basename(path)[12:-3], len(arr), acc["fast_cells"], acc["pyramid"], acc["quadtree"], acc["blur"], acc["orient_brief"]), flush=True)
    ex.close()
