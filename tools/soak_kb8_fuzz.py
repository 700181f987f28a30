"""Fisheye stereo (Frame::ComputeStereoFishEyeMatches -> KannalaBrandt8::TriangulateMatches) with RANDOM cameras: focal lengths, principal points, the four
Kannala-Brandt coefficients, the left-to-right rotation and baseline - product vs the reference's own Frame.cc + KannalaBrandt8.cpp (oracle/_ref), match sets
identical, mvDepth / mvStereo3Dpoints identical to the bit.  tests/test_kb8.py and tools/soak_round5.py use the TUM-VI rig only.
    python tools/soak_kb8_fuzz.py hip|emu FIRST LAST"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from orb_slam3_detailed_comments_amd import _lib, synth, sophus
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
from orb_slam3_detailed_comments_amd import matcher as M
from matcher_world import rot

kind, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = _lib.load_hip() if kind == "hip" else _lib.OrbxLib(os.environ.get("ORBX_SOAK_LIB") or os.path.join(ROOT, "tests", "emu", "liborbx_emu.so"))
assert ol.reference_frame_lib() is not None, "oracle/_ref/libref_frame.so not built (needs /root/reference)"
bad = pairs = accepted = rejected = 0
for seed in range(first, last + 1):
    rng = np.random.default_rng(9000 + seed)
    w = h = 512
    def cam():
        f = float(rng.uniform(120.0, 420.0))
        return [f, f * float(rng.uniform(0.97, 1.03)), w / 2 + float(rng.uniform(-20, 20)), h / 2 + float(rng.uniform(-20, 20))] + \
               [float(rng.normal(0, s)) for s in (0.03, 0.01, 0.005, 0.001)]
    c1, c2 = cam(), cam()
    c2[0] = c1[0] * float(rng.uniform(0.98, 1.02)); c2[1] = c1[1] * float(rng.uniform(0.98, 1.02))
    R = rot(*rng.normal(0, 0.02, 3)).astype(np.float32)
    t = np.array([float(rng.uniform(0.04, 0.3)), float(rng.normal(0, 0.003)), float(rng.normal(0, 0.003))], np.float32)
    nf = int(rng.integers(500, 1800)); lap = (0, 511) if rng.integers(0, 2) else (int(rng.integers(0, 150)), int(rng.integers(350, 511)))
    L, Rt = synth.stereo_pair(w, h, seed=seed, nrect=int(rng.integers(800, 2500)), max_disp=int(rng.integers(6, 40)), band=int(rng.integers(16, 96)))
    F = ol.reference_fisheye_frame(L, Rt, lap, lap, nf, cams=(c1, c2, R, t))
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    (mL, kL, dL), (mR, kR, dR) = ex.extract_batch(np.stack([L, Rt]), lap)
    out = M.ComputeStereoFishEyeMatches(ex, ex, c1, c2, sophus.SE3f(R, t).rotationMatrix(), t, 0, 1, 1)
    nl, nr = len(kL), len(kR)
    acc = F["l2r"] >= 0
    ok = kL.tobytes() == F["keys"].tobytes() and np.array_equal(out["l2r"][0, :nl], F["l2r"]) and np.array_equal(out["r2l"][0, :nr], F["r2l"]) and \
        out["depth"][0, :nl][acc].tobytes() == F["depth"][acc].tobytes() and out["p3d"][0, :nl][acc].tobytes() == F["p3d"][acc].tobytes() and \
        np.all(out["depth"][0, :nl][~acc] == -1.0)
    pairs += 1; accepted += int(acc.sum()); rejected += int((~acc).sum())
    if not ok:
        bad += 1
        print("seed %d DIFFERS: cams %s %s t %s" % (seed, np.round(c1, 4).tolist(), np.round(c2, 4).tolist(), t.tolist()), flush=True)
    ex.close()
    if (seed - first) % 25 == 24:
        print("seeds %d..%d: %d pairs, %d accepted matches, %d differences so far" % (first, seed, pairs, accepted, bad), flush=True)
print("Kannala-Brandt camera fuzz (%s library vs the reference Frame.cc + KannalaBrandt8.cpp): seeds %d..%d, %d pairs, %d accepted / %d rejected matches, %d differences"
      % (kind, first, last, pairs, accepted, rejected, bad))
