"""Soak of tests/test_emu_search.run_all (every matcher method - GetFeaturesInArea, both SearchByProjection overloads, SearchForTriangulation single / batched /
resident, SearchByBoW x2 single / batched / resident, SearchForInitialization, the Sim3 / key-frame / Fuse / SearchBySim3 variants - on one scene per seed)
beyond the seeds of the suite, product vs the oracle restatement:   python tools/soak_search_scenes.py hip|emu FIRST LAST"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from orb_slam3_detailed_comments_amd import _lib
import test_emu_search as t

kind, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = _lib.load_hip() if kind == "hip" else _lib.OrbxLib(os.environ.get("ORBX_SOAK_LIB") or os.path.join(ROOT, "tests", "emu", "liborbx_emu.so"))
bad = 0
for s in range(first, last + 1):
    try:
        t.run_all(lib, 376, 240, 500, 300, [s])
    except AssertionError as e:
        bad += 1
        print("seed %d DIFFERS: %s" % (s, str(e)[:200]), flush=True)
    if (s - first) % 20 == 19:
        print("seeds %d..%d: %d differences so far" % (first, s, bad), flush=True)
print("search scenes soak (%s library): seeds %d..%d, %d differences" % (kind, first, last, bad))
