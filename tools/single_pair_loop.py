"""One stereo pair per call, synchronised after every pair (Tracking's rhythm), frames written into pyramid level 0 by the producer:
the loop that tests/gpu_quick.py times, alone, for a rocprofv3 kernel trace of the launch chain.

    python tools/single_pair_loop.py [N] [forms]     forms: 1 (default) the small-batch launch form, 0 the large-batch form, ab = both, alternating
Also times one image per call (the monocular rhythm: extract, wait)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam3_detailed_comments_amd import synth, load_hip, _lib
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
lib = _lib.OrbxLib(os.environ['ORBX_QUICK_LIB']) if os.environ.get('ORBX_QUICK_LIB') else load_hip()      # ORBX_QUICK_LIB: A/B of library builds
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mode = sys.argv[2] if len(sys.argv) > 2 else "1"
L, R = synth.stereo_pair(seed=100)
pair = np.stack([L, R])
bf, b = 458.654 * 0.110074, 0.110074


def run(forms, n):
    ex = ORBextractor(1200, 1.2, 8, 20, 7, lib=lib)
    ex.set_small_batch_forms(forms)
    ptr, shp, strd, istrd = ex.input_upload(pair)

    def one_pair():
        ex.enqueue(None, (0, 0), device_ptr=ptr, shape=shp, stride=strd, image_stride=istrd); lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 1, 1, bf, b)); ex.sync()
    for it in range(20):
        one_pair()
    t = time.perf_counter()
    for it in range(n):
        one_pair()
    pair_ms = (time.perf_counter() - t) / n * 1e3
    ex.close()
    ex = ORBextractor(1000, 1.2, 8, 20, 7, lib=lib)
    ex.set_small_batch_forms(forms)
    ptr, shp, strd, istrd = ex.input_upload(L[None])
    for it in range(20):
        ex.enqueue(None, (0, 0), device_ptr=ptr, shape=shp, stride=strd, image_stride=istrd); ex.sync()
    t = time.perf_counter()
    for it in range(n):
        ex.enqueue(None, (0, 0), device_ptr=ptr, shape=shp, stride=strd, image_stride=istrd); ex.sync()
    mono_ms = (time.perf_counter() - t) / n * 1e3
    ex.close()
    # throughput of small batches on one handle (back-to-back calls, one wait at the end): the forms must not cost anything here
    rates = []
    for P in (4, 16):
        Ls, Rs = zip(*[synth.stereo_pair(seed=100 + s) for s in range(P)])
        arr = np.stack(list(Ls) + list(Rs))
        ex = ORBextractor(1200, 1.2, 8, 20, 7, lib=lib)
        ex.set_small_batch_forms(forms)
        ptr, shp, strd, istrd = ex.input_upload(arr)
        for it in range(3):
            ex.enqueue(None, (0, 0), device_ptr=ptr, shape=shp, stride=strd, image_stride=istrd); lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, P, P, bf, b))
        ex.sync()
        K = 30
        t = time.perf_counter()
        for it in range(K):
            ex.enqueue(None, (0, 0), device_ptr=ptr, shape=shp, stride=strd, image_stride=istrd); lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, P, P, bf, b))
        ex.sync()
        rates.append("%d pairs/call: %.0f pairs/s" % (P, P * K / (time.perf_counter() - t)))
        ex.close()
    return pair_ms, mono_ms, rates


if mode == "ab":
    for rep in range(3):
        for forms in (1, 0):
            p, m, rates = run(forms, N)
            print("small-batch forms %s: single pair %.4f ms, single image %.4f ms (%d calls each, zero-copy input, sync after every call); %s" % (forms, p, m, N, "; ".join(rates)), flush=True)
else:
    p, m, _ = run(int(mode), N)
    print("single pair, zero-copy input: %.3f ms per pair over %d pairs; single image: %.3f ms" % (p, N, m))
