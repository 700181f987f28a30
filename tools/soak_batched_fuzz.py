"""The batched device searches of rounds 3-4 with RANDOM parameters (the suites run two or three fixed sets): TrackWithMotionModel's search per frame (th, occupied
keypoints, orientation check, mono / stereo), the relocalisation search (th, ORBdist), SearchByBoW(KeyFrame, Frame) for a batch (nnratio), Tracking::SearchLocalPoints for a batch (th, far points, viewing cosine, nnratio) - each frame against
the reference's own Frame.cc / ORBmatcher.cc.   python tools/soak_batched_fuzz.py hip|emu FIRST LAST"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from orb_slam3_detailed_comments_amd import _lib
import test_lastframe_batch as t_last, test_keyframe_batch as t_kf, test_bow_frames_batch as t_bow, test_local_points_batch as t_lp

kind, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = _lib.load_hip() if kind == "hip" else _lib.OrbxLib(os.environ.get("ORBX_SOAK_LIB") or os.path.join(ROOT, "tests", "emu", "liborbx_emu.so"))
for m in (t_last, t_kf, t_bow, t_lp): m.STRICT_SCENES = False
bad = runs = 0
for seed in range(first, last + 1):
    rng = np.random.default_rng(17000 + seed)
    t_last.PARAM_SETS = tuple((float(rng.choice([1.0, 2.0, 4.0, 7.0, 15.0, 30.0])), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))) for _ in range(2))
    t_kf.PARAM_SETS = tuple((float(rng.choice([1.0, 3.0, 6.0, 10.0, 25.0])), int(rng.choice([30, 50, 64, 100, 160])), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))) for _ in range(2))
    t_bow.PARAM_SETS = tuple((float(rng.uniform(0.5, 1.0)), bool(rng.integers(0, 2))) for _ in range(2))
    t_lp.PARAM_SETS = tuple((float(rng.choice([0.5, 1.0, 2.0, 3.0, 6.0, 15.0])), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), float(rng.choice([0.5, 0.0, 0.8])),
                             float(rng.uniform(2.0, 30.0)), float(rng.uniform(0.5, 1.0))) for _ in range(2))
    for name, f in (("lastframe", lambda: t_last._run(lib, 376, 240, 500, 3, bool(seed & 1), seed)), ("keyframe", lambda: t_kf._run(lib, 376, 240, 500, 3, seed)),
                    ("bow", lambda: t_bow._run(lib, 376, 240, 500, 3, seed)), ("local_points", lambda: t_lp._run(lib, 376, 240, 500, 3, 900, bool(seed & 2)))):
        runs += 1
        try:
            f()
        except AssertionError as e:
            bad += 1
            print("seed %d %s DIFFERS: %s  params %s" % (seed, name, str(e)[:160], {"lastframe": t_last.PARAM_SETS, "keyframe": t_kf.PARAM_SETS, "bow": t_bow.PARAM_SETS, "local_points": t_lp.PARAM_SETS}[name]), flush=True)
    if (seed - first) % 10 == 9:
        print("seeds %d..%d: %d runs, %d differences so far" % (first, seed, runs, bad), flush=True)
print("batched searches parameter fuzz (%s library vs the reference): seeds %d..%d, %d runs, %d differences" % (kind, first, last, runs, bad))
