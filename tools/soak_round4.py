"""Soak of the round-4 paths on a GPU box, many seeds each, against what the tests of the same name compare with - the reference's own Frame.cc /
ORBmatcher.cc per frame: (a) the batched LastFrame search (quaternion pose action), (b) the batched relocalisation search
SearchByProjection(Frame, KeyFrame, sAlreadyFound, th, ORBdist), (c) the batched SearchByBoW(KeyFrame, Frame), (d) map points constructed on the
edges of the search windows (product == reference; the matrix form of Tcw * p caught on every frame).
    python tools/soak_round4.py [seeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from orb_slam3_detailed_comments_amd import _lib             # noqa: E402
import test_lastframe_batch as t_last                        # noqa: E402
import test_keyframe_batch as t_kf                           # noqa: E402
import test_bow_frames_batch as t_bow                        # noqa: E402
import test_sophus_action as t_so3                           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
LIB_PATH = os.environ.get("ORBX_SOAK_LIB") or _lib.HIP_LIB_PATH      # ORBX_SOAK_LIB=tests/emu/liborbx_emu.so: the same soak on the CPU emulator build
lib = _lib.OrbxLib(LIB_PATH) if os.environ.get("ORBX_SOAK_LIB") else _lib.load_hip()
for s in range(1, n + 1):
    t_last._run(lib, 752, 480, 1200, 6, bool(s & 1), seed=s)
print("batched LastFrame search vs the reference Frame + ORBmatcher.cc per frame: %d batches of 6 frames x 3 settings, 0 differences" % n, flush=True)
for s in range(1, n + 1):
    t_kf._run(lib, [752, 640][s & 1], 480, [1200, 1000][s & 1], 6, seed=s)
print("batched relocalisation search vs the reference per frame: %d batches of 6 frames x 3 settings, 0 differences" % n, flush=True)
for s in range(1, n + 1):
    t_bow._run(lib, 752, 480, 1200, 4, seed=s)
print("batched SearchByBoW(KeyFrame, Frame) vs the reference ORBmatcher.cc per frame: %d batches of 4 frames x 2 settings, 0 differences" % n, flush=True)
for s in range(1, max(2, n // 2) + 1):
    t_so3._run(lib, 752, 480, 1200, 3, seed=s)
print("edge-of-window map points: product == reference, matrix form of Tcw * p caught, %d runs of 3 frames" % max(2, n // 2), flush=True)
