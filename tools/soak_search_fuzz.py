"""Soak of tests/test_emu_search_fuzz.py's random scenes (every projection-type search, GetFeaturesInArea, the window search with random radii, levels and
occupancy) beyond the six seeds of the suite:   python tools/soak_search_fuzz.py hip|emu FIRST LAST"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from orb_slam3_detailed_comments_amd import _lib
import test_emu_search_fuzz as t

kind, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = _lib.load_hip() if kind == "hip" else _lib.OrbxLib(os.environ.get("ORBX_SOAK_LIB") or os.path.join(ROOT, "tests", "emu", "liborbx_emu.so"))
bad = 0
for seed in range(first, last + 1):
    try:
        t.test_projection_searches_fuzz(lib, seed)
    except AssertionError as e:
        bad += 1
        print("seed %d DIFFERS: %s" % (seed, str(e)[:300]), flush=True)
    if (seed - first) % 50 == 49:
        print("seeds %d..%d: %d differences so far" % (first, seed, bad), flush=True)
print("search fuzz soak (%s library): seeds %d..%d, %d differences" % (kind, first, last, bad))
