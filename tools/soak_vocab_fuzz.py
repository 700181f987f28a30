"""ORBVocabulary::transform with RANDOM vocabularies - branching factor 2-12, depth 1-6, both scoring types, all four weighting types, ragged trees (leaves at
several depths), levelsup 0..L+1, feature counts 0-2000 - product vs the reference's own Thirdparty/DBoW2 (oracle/_ref/libref_dbow2.so): BowVector ids and
values (doubles, bit-exact), FeatureVector nodes and feature lists.  The suites run seven fixed configurations.
    python tools/soak_vocab_fuzz.py hip|emu FIRST LAST"""
import os, pathlib, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from orb_slam3_detailed_comments_amd import _lib
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
import test_emu_vocab as t

kind, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = _lib.load_hip() if kind == "hip" else _lib.OrbxLib(os.environ.get("ORBX_SOAK_LIB") or os.path.join(ROOT, "tests", "emu", "liborbx_emu.so"))
ex = ORBextractor(500, 1.2, 8, 20, 7, lib=lib)
tmp = pathlib.Path(tempfile.mkdtemp())
bad = 0
for seed in range(first, last + 1):
    rng = np.random.default_rng(11000 + seed)
    k = int(rng.integers(2, 13)); L = int(rng.integers(1, 7))
    while k ** L > 60000: L -= 1
    ragged = bool(rng.integers(0, 2)) and L >= 2
    mll = int(rng.integers(1, max(2, L))) if ragged else 1
    # levelsup: the FeatureVector node is taken at level L - levelsup.  A leaf ABOVE that level leaves the reference's `nid` an uninitialised local
    # (TemplatedVocabulary.h:1150: undefined; ORBvoc at levelsup 4 has no leaf above level 2), so ragged trees keep L - levelsup <= their first leaf level
    lo = max(0, L - mll) if ragged else 0
    cfg = (k, L, int(rng.choice([0, 1, 2, 5])), int(rng.integers(0, 4)), ragged, mll, tuple(sorted(set(int(x) for x in rng.integers(lo, L + 2, 3)))))
    try:
        t.check_vocabulary(ex, tmp, cfg, seed=seed, n_desc=int(rng.integers(1, 2000)))
    except AssertionError as e:
        bad += 1
        print("seed %d DIFFERS: cfg %s: %s" % (seed, cfg, str(e)[:200]), flush=True)
    if (seed - first) % 20 == 19:
        print("seeds %d..%d: %d differences so far" % (first, seed, bad), flush=True)
print("vocabulary fuzz (%s library vs the reference DBoW2): seeds %d..%d, %d differences" % (kind, first, last, bad))
