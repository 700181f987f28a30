import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from orb_slam3_detailed_comments_amd import _lib
import test_lastframe_batch as t_last, test_keyframe_batch as t_kf, test_bow_frames_batch as t_bow, test_sophus_action as t_so3
lib = _lib.OrbxLib("/root/repo/tests/emu/liborbx_emu.so")
n = int(sys.argv[1])
for s in range(1, n + 1):
    t_last._run(lib, 376, 240, 500, 3, bool(s & 1), seed=s)
    t_kf._run(lib, [376, 320][s & 1], 240, [500, 400][s & 1], 3, seed=s)
    t_bow._run(lib, 376, 240, 500, 2, seed=s)
    t_so3._run(lib, 376, 240, 500, 2, seed=s)
    print("seed", s, "ok", flush=True)
print("emulator soak of the round-4 families (batched LastFrame / relocalisation / SearchByBoW searches vs the reference per frame, edge-of-window points): %d seeds x 4 families, 0 differences" % n)
