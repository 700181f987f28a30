#!/bin/bash
# A/B of two library builds (build/variants/liborbx_hip_prev.so, liborbx_hip_new.so): serial stage times, then the bench line twice each.
mkdir -p gpurun_out/r02
python tools/time_fast_variants.py prev new 2>&1 | grep "B=" | tee gpurun_out/r02/ab.txt
python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from orb_slam3_detailed_comments_amd import _lib, synth
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
from orb_slam3_detailed_comments_amd import matcher as M
for name in ("prev", "new"):
    lib = _lib.OrbxLib("build/variants/liborbx_hip_%s.so" % name)
    ex = ORBextractor(1200, 1.2, 8, 20, 7, lib=lib)
    imgs = []
    for s in range(64):
        l, r = synth.stereo_pair(seed=100 + s); imgs.append((l, r))
    arr = np.stack([p[0] for p in imgs] + [p[1] for p in imgs])
    d = ex.device_upload(arr); ex.profile(True, serial=True)
    acc = 0.0
    for it in range(13):
        ex.enqueue(None, (0, 0), device_ptr=d, shape=arr.shape); lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 64, 64, 458.654 * 0.110074, 0.110074)); ex.sync()
        u, dd, n = M.ComputeStereoMatches(ex, ex, 458.654 * 0.110074, 0.110074, 0, 64, 64)
        if it >= 3: acc += ex.stage_ms()["match"] / 10
    print("%s: stereo match stage, 64 pairs: %.4f ms" % (name, acc))
PY
for v in prev new prev new; do echo -n "$v "; ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | cut -c125-160; done
