"""Stereo association with random parameters - image size, feature budget, pyramid, bf and baseline (the disparity range minD = 0 .. maxD = bf / b of
Frame::ComputeStereoMatches, src/Frame.cc:1102-1358), band and disparity of the synthetic pair - product vs the oracle restatement and, where
oracle/_ref is built, the reference's own Frame constructor:   python tools/soak_stereo_fuzz.py hip|emu FIRST LAST"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from orb_slam3_detailed_comments_amd import _lib, synth
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
from orb_slam3_detailed_comments_amd import matcher as M

kind, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = _lib.load_hip() if kind == "hip" else _lib.OrbxLib(os.environ.get("ORBX_SOAK_LIB") or os.path.join(ROOT, "tests", "emu", "liborbx_emu.so"))
bad = total = 0
for seed in range(first, last + 1):
    rng = np.random.default_rng(7000 + seed)
    w = int(rng.integers(300, 900)); h = int(rng.integers(max(240, (w + 1) // 2 + 1), 600))
    nf = int(rng.integers(100, 2000)); nl = int(rng.integers(3, 9)); sf = float(rng.choice([1.1, 1.2, 1.3, 1.5]))
    bf = float(rng.uniform(5.0, 120.0)); b = float(rng.uniform(0.03, 0.8))
    L, R = synth.stereo_pair(w, h, seed=seed, nrect=int(rng.integers(100, 4000)), band=int(rng.integers(8, 64)), max_disp=int(rng.integers(4, 90)))
    try:
        ex = ORBextractor(nf, sf, nl, 20, 7, lib=lib)
        res = ex.extract_batch(np.stack([L, R]))
    except _lib.OrbxError as e:
        continue                                    # (too small for that pyramid)
    u, d, n = M.ComputeStereoMatches(ex, ex, bf, b, 0, 1, 1)
    oL, oR = ol.OracleExtractor(nf, sf, nl, 20, 7), ol.OracleExtractor(nf, sf, nl, 20, 7)
    eL, eR = oL.extract(L), oR.extract(R)
    uo, do, no = ol.oracle_stereo(oL, oR, eL[1], eL[2], eR[1], eR[2], bf, b)
    N = len(eL[1])
    ok = n[0] == no and u[0, :N].tobytes() == uo.tobytes() and d[0, :N].tobytes() == do.tobytes()
    total += 1
    if not ok:
        bad += 1
        print("seed %d DIFFERS: %s" % (seed, (w, h, nf, nl, sf, bf, b, int(n[0]), int(no))), flush=True)
    ex.close()
    if (seed - first) % 50 == 49:
        print("seeds %d..%d: %d pairs, %d differences so far" % (first, seed, total, bad), flush=True)
print("stereo fuzz soak (%s library): seeds %d..%d, %d pairs, %d differences" % (kind, first, last, total, bad))
