import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from orb_slam3_detailed_comments_amd import synth, _lib, ORBextractor
lib = _lib.OrbxLib(os.path.join(ROOT, 'build', 'variants', 'liborbx_hip_fastspan.so'))
L, R = synth.stereo_pair(seed=100)
pair = np.stack([L, R])
ex = ORBextractor(1200, 1.2, 8, 20, 7, lib=lib)
dptr = ex.device_upload(pair)
for it in range(3):
    print("--- run", it, flush=True)
    ex.enqueue(None, (0, 0), device_ptr=dptr, shape=pair.shape); ex.sync()
