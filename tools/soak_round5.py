"""Soak of the round-5 change on a GPU box: every Kannala-Brandt path of the product against the reference's OWN src/CameraModels/KannalaBrandt8.cpp
(compiled unmodified into oracle/_ref, round 5), many seeds each, bit for bit:
 (a) the fisheye-rig Frame constructor: mvLeftToRightMatch / mvRightToLeftMatch / mvDepth / mvStereo3Dpoints (tests/test_kb8.py),
 (b) isInFrustum + SearchByProjection over the two-camera rig (tests/test_local_points_rig.py),
 (c) SearchForTriangulation on Kannala-Brandt key frames, one camera and the rig, stand-in and real-class worlds through the facade's implicit
     resident key-frame cache (tests/test_matcher_reference.py, variant "kb8"),
 (d) the batched LastFrame / relocalisation searches on frames of one Kannala-Brandt camera (tests/test_lastframe_batch.py).
    python tools/soak_round5.py [seeds]"""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from orb_slam3_detailed_comments_amd import _lib             # noqa: E402
import test_kb8 as t_kb8                                     # noqa: E402
import test_local_points_rig as t_rig                        # noqa: E402
import test_matcher_reference as t_mw                        # noqa: E402
import test_lastframe_batch as t_last                        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LIB_PATH = os.environ.get("ORBX_SOAK_LIB") or _lib.HIP_LIB_PATH      # ORBX_SOAK_LIB=tests/emu/liborbx_emu.so: the same soak on the CPU emulator build
lib = _lib.OrbxLib(LIB_PATH) if os.environ.get("ORBX_SOAK_LIB") else _lib.load_hip()
acc = t_kb8._check(lib, tuple(range(100, 100 + n)), (0, 511), 1500) + t_kb8._check(lib, tuple(range(200, 200 + max(1, n // 2))), (100, 400), 1000)
print("fisheye-rig Frame constructor vs the reference Frame.cc + KannalaBrandt8.cpp: %d pairs, %d accepted matches, mvDepth / mvStereo3Dpoints identical to the bit" % (n + max(1, n // 2), acc), flush=True)
t_rig._check(lib, tuple(range(300, 300 + max(2, n // 2))), 3000)
print("isInFrustum + SearchByProjection over the rig vs the reference: %d frames x 2 settings, 0 differences" % max(2, n // 2), flush=True)
tmp = pathlib.Path(tempfile.mkdtemp())
t_mw._compare(tmp, LIB_PATH, [(s, "kb8") for s in range(400, 400 + n)])
t_mw._compare(tmp, LIB_PATH, [(s, "kb8") for s in range(500, 500 + max(2, n // 2))] + [(s, "base") for s in range(600, 600 + max(2, n // 2))], t_mw.REF_REAL, t_mw.FACADE_REAL)
print("matcher worlds (Kannala-Brandt key frames, one camera and rig; single calls through the implicit resident cache) vs the reference ORBmatcher.cc: %d stand-in + %d real-class worlds, 0 differences"
      % (n, 2 * max(2, n // 2)), flush=True)
for s in range(max(2, n // 2)):
    t_last._kb8_case(lib, 512, 512, 1500, 3 + s)
print("batched LastFrame / relocalisation searches on Kannala-Brandt frames vs the reference: %d runs, 0 differences" % max(2, n // 2), flush=True)
