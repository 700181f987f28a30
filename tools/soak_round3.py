"""Soak of the round-3 paths on a GPU box: many seeds of (a) the rig / Kannala-Brandt isInFrustum + rig SearchByProjection, (b) the batched
SearchLocalPoints on stereo and RGB-D frames, (c) undistorted RGB-D frames, (d) the batched LastFrame search - each against what the tests of the
same name compare with (the reference's own Frame.cc / ORBmatcher.cc, or the single-frame product call that is pinned to it).
    python tools/soak_round3.py [seeds]
Prints one line per family; exits non-zero on the first difference (the test helpers assert)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                           # noqa: E402
from orb_slam3_detailed_comments_amd import _lib             # noqa: E402
import test_local_points_rig as t_rig                        # noqa: E402
import test_local_points_batch as t_batch                    # noqa: E402
import test_undistort as t_und                               # noqa: E402
import test_lastframe_batch as t_last                        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
LIB_PATH = os.environ.get("ORBX_SOAK_LIB") or _lib.HIP_LIB_PATH      # ORBX_SOAK_LIB=tests/emu/liborbx_emu.so: the same soak on the CPU emulator build
lib = _lib.OrbxLib(LIB_PATH) if os.environ.get("ORBX_SOAK_LIB") else _lib.load_hip()
t_rig._check(lib, tuple(range(200, 200 + n)), 5000)
print("rig isInFrustum + rig SearchByProjection vs the reference rig Frame: %d frames x 5000 points x 2 settings, 0 differences" % n, flush=True)
for i in range(max(1, n // 3)):
    for rgbd in (False, True):
        t_batch._run(lib, [640, 752][i % 2], 480, [1000, 1200][i % 2], 3 - (i % 2), 3000 + 500 * i, rgbd)      # the helper has three poses
print("batched SearchLocalPoints vs the reference Frame per frame: %d batches (stereo + RGB-D), 0 differences" % (2 * max(1, n // 3)), flush=True)
for i in range(max(1, n // 4)):
    t_batch._large_batch(lib, 640, 480, 1000, 32 + 16 * (i % 2), 5000, 8 + i)
print("large batches vs the single-frame call: %d, 0 differences" % max(1, n // 4), flush=True)
for i in range(max(1, n // 3)):
    t_und._product_case(lib, 640, 480, 1000, 3 + (i % 3), 3000, True)
print("undistorted RGB-D frames vs the reference Frame (TUM1 coefficients): %d batches, 0 differences" % max(1, n // 3), flush=True)
for i in range(max(1, n // 3)):
    t_last._run(lib, [752, 640][i % 2], 480, [1200, 1000][i % 2], 8 + 8 * (i % 3), bool(i & 1))
print("batched LastFrame search vs the single-frame call: %d batches, 0 differences" % max(1, n // 3), flush=True)
