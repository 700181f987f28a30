"""orbx_set_host_wait after PyTorch has initialised the device (the order of bench.py --gpus N with the RCCL backend): must succeed and must take effect."""
import os, sys, time
sys.path.insert(0, ".")
import torch
torch.cuda.set_device(0); x = torch.zeros(8, device="cuda"); torch.cuda.synchronize()
import numpy as np
from orb_slam3_detailed_comments_amd import ORBextractor, load_hip, synth
lib = load_hip()
rc = lib.L.orbx_set_host_wait(0, 1)
print("orbx_set_host_wait after torch init: rc", rc, (lib.L.orbx_last_error() or b"").decode() if rc else "")
ex = ORBextractor(1200, 1.2, 8, 20, 7, device_id=0, lib=lib)
batch = np.stack([synth.corner_field(seed=i) for i in range(64)])
lp, shape, stride, istride = ex.input_upload(batch)
for _ in range(3):
    ex.enqueue(None, (0, 0), device_ptr=lp, shape=shape, stride=stride, image_stride=istride); ex.sync()
c0 = time.thread_time(); t0 = time.time()
for _ in range(300):
    ex.enqueue(None, (0, 0), device_ptr=lp, shape=shape, stride=stride, image_stride=istride); ex.sync()
print("calling thread: %.2f s CPU over %.2f s wall (blocking waits leave most of the wall time idle)" % (time.thread_time() - c0, time.time() - t0))
