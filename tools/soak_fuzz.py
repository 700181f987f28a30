"""Soak of tests/test_emu_fuzz.py's random configurations (image size, pyramid depth and factor 1.1-2.0, thresholds, feature budget 1-1000, four image
families, lapping areas, both Gaussian tap sets) beyond the 14 seeds of the suite, against the oracle AND the reference build:
    python tools/soak_fuzz.py hip|emu FIRST LAST [pool]
Alternates the pyramid launch forms and the small-batch / large-batch launch forms of the extraction.  With `pool` (round 6) the feature budget is raised
(300 .. 20 000) and the quadtree's node-pool form is forced on a varying share of the levels (orbx_debug_quadtree_lds_nodes: 0 / 20 / 150 / the default)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from orb_slam3_detailed_comments_amd import _lib
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
from test_emu_fuzz import _case

kind, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
POOL = len(sys.argv) > 4 and sys.argv[4] == "pool"
lib = _lib.load_hip() if kind == "hip" else _lib.OrbxLib(os.path.join(ROOT, "tests", "emu", "liborbx_emu.so"))
bad = rejected = 0
for seed in range(first, last + 1):
    img, nf, sf, nl, ini, mn, lap, gv = _case(seed)
    if POOL:
        nf = int(np.random.default_rng(77000 + seed).choice([300, 1500, 6000, 20000]))
    try:
        ex = ORBextractor(nf, sf, nl, ini, mn, lib=lib)
        if POOL:
            ex.debug_quadtree_lds_nodes((0, 20, 150, 4000)[seed % 4])
        ex.set_gaussian_taps(gv)
        ex.pyramid_mode(1 + seed % 2)
        ex.set_small_batch_forms(bool((seed >> 1) & 1))
        got = ex(img, None, lap)
        ex.close()
    except _lib.OrbxError as e:
        print("seed %d library error %s" % (seed, e), flush=True); rejected += 1
        continue
    exp = ol.OracleExtractor(nf, sf, nl, ini, mn, gv).extract(img, lap)
    ok = got[0] == exp[0] and ol.kps_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2])
    if ol.reference() is not None:
        ref = ol.ReferenceExtractor(nf, sf, nl, ini, mn, gv).extract(img, lap)
        ok = ok and got[0] == ref[0] and ol.kps_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    if not ok:
        bad += 1
        print("seed %d DIFFERS: %s" % (seed, (img.shape, nf, sf, nl, ini, mn, lap, gv)), flush=True)
    if (seed - first) % 50 == 49:
        print("seeds %d..%d: %d differences so far" % (first, seed, bad), flush=True)
print("fuzz soak%s (%s library, oracle + reference build%s): seeds %d..%d, %d configurations rejected by the library (image larger than 4127 px), %d differences"
      % (", quadtree node pool forced" if POOL else "", kind, "" if ol.reference() is not None else " ABSENT", first, last, rejected, bad))
