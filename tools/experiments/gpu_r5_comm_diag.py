"""Which step of tests/test_lifetime.exercise leaves a HIP error behind (RCCL's init then reports "unhandled cuda error")?"""
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from orb_slam3_detailed_comments_amd import _lib, multi, synth
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
from orb_slam3_detailed_comments_amd import matcher as M
hip = C.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = C.c_char_p
lib = _lib.load_hip()
def peek(tag):
    e = hip.hipPeekAtLastError()
    print("%-28s hipPeekAtLastError = %d %s" % (tag, e, hip.hipGetErrorString(e).decode() if e else ""), flush=True)
    if e: hip.hipGetLastError()
def try_comm(tag):
    try:
        c = multi.Communicator(lib, 1, 0, multi.Communicator.unique_id(lib), device_id=0); c.close(); print(tag, "communicator ok", flush=True)
    except Exception as ex:
        print(tag, "communicator FAILED:", str(ex)[:200], flush=True); peek("after failed comm")
import test_bow_frames_batch as t_bow, test_emu_mappoint as t_mp, test_emu_search as t_search, test_keyframe_batch as t_kf, test_lastframe_batch as t_lf, test_local_points_batch as t_lp
w, h, nf = 752, 480, 1200
peek("start"); try_comm("start")
for name, f in [("bow", lambda: t_bow._run(lib, w, h, nf, 3)), ("kf", lambda: t_kf._run(lib, w, h, nf, 3)), ("lf", lambda: t_lf._run(lib, w, h, nf, 3, False)),
                ("lp rgbd", lambda: t_lp._run(lib, w, h, nf, 3, 900, True)), ("lp", lambda: t_lp._run(lib, w, h, nf, 3, 900, False)), ("mp", lambda: t_mp.run(lib, 40, 60, 1)),
                ("search", lambda: t_search.run_all(lib, w, h, nf, 300, [0]))]:
    f(); peek(name); try_comm(name)
exs = [ORBextractor(nf, 1.2, 8, 20, 7, lib=lib) for _ in range(3)]
imgs = np.stack([synth.corner_field(w, h, seed=s) for s in range(4)])
for i, ex in enumerate(exs):
    lib.check(lib.L.orbx_set_graph_replay(ex._h, i & 1)); peek("graph flag %d" % i)
    ex.extract_batch(imgs[:2 + i % 2]); peek("extract a %d" % i)
    ex.extract_batch(imgs); peek("extract b %d" % i)
    lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 2, 2, 40.0, 0.1)); M.StereoFetch(ex, 2); peek("stereo %d" % i)
    M.StereoFishEyeKnn(ex, ex, 0, 2, 2); peek("knn %d" % i)
    try_comm("handle %d" % i)
ex = exs[0]
yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
ex.set_input(3, True, 0, remap=(xx + 0.25, yy - 0.5)); ex.extract_batch(np.repeat(imgs[:2, :, :, None], 3, axis=3)); peek("remap")
ex.set_input(1, True, 0, resize=(w, h)); ex.extract_batch(np.repeat(np.repeat(imgs[:2], 2, axis=1), 2, axis=2)); peek("resize")
ex.set_input(None)
ex.set_undistort((300.0, 300.0, w / 2, h / 2), (0.1, -0.05, 1e-3, 1e-3, 0.01)); ex.profile(True, serial=True)
r = ex.extract_batch(imgs[:2]); ex.fetch_undistorted(); peek("undistort + serial profile")
prof = (C.c_longlong * 16)(); lib.L.orbx_debug_quadtree_profile(ex._h, prof); peek("quadtree probe")
ex.profile(False); ex.set_undistort(None)
M.ORBmatcher.DescriptorDistance(ex, r[0][2][:50], r[1][2][:70]); peek("hamming")
dp, hp = C.c_void_p(), C.c_void_p()
lib.check(lib.L.orbx_device_alloc(ex._h, 1 << 20, C.byref(dp))); lib.check(lib.L.orbx_host_alloc(ex._h, 1 << 20, C.byref(hp))); peek("caller buffers")
try_comm("end")
