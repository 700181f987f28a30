"""Builds build/variants/liborbx_hip_<name>.so for a few compile-time tunables (GPU sweep: tools/experiments/gpu_sweep_tunables.sh)."""
import os, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from build_fast_variants import build, OUT
JOBS = [("base+ref", ()), ("base+blur24", ("-DORBX_BLUR_ROWS=24",)), ("base+blur32", ("-DORBX_BLUR_ROWS=32",)), ("base+strip8", ("-DORBX_RESIZE_STRIP=8",)), ("base+strip32", ("-DORBX_RESIZE_STRIP=32",)),
        ("base+list1k", ("-DORBX_FAST_LIST_BYTES=1024",)), ("base+list1536", ("-DORBX_FAST_LIST_BYTES=1536",))] + [("base+rs%d" % k, ("-DORBX_STEREO_ROW_SHIFT=%d" % k,)) for k in (2, 4, 5)]
JOBS += [("base+pt%d_%d" % (t, ts), ("-DORBX_PYR_THREADS=%d" % t, "-DORBX_PYR_TILE=%d" % ts)) for t in (256, 512, 1024) for ts in (8, 16, 32)]
if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    jobs = [j for j in JOBS if len(sys.argv) == 1 or j[0] in sys.argv[1:]]
    with ThreadPoolExecutor(4) as ex:
        for o in ex.map(lambda j: build(*j), jobs): print(o)
