"""Lifetime of everything the library allocates: extractor handles (streams, events, device and page-locked buffers, the captured graph),
device-resident key frames and map points, vocabularies, communicators and caller-owned buffers are created, used and destroyed, and the
library's own count of what it holds (orbx_debug_live_resources) has to come back to where it started - on the emulator build here, on the
HIP build (plus the device's free memory) with -m gpu.  The work in between is the parity runners of the other test modules, so every
allocation path those reach is covered."""
import ctypes as C
import gc

import numpy as np
import pytest

from orb_slam3_detailed_comments_amd import multi, synth
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
from orb_slam3_detailed_comments_amd import matcher as M


def live(lib):
    gc.collect()
    a = (C.c_longlong * 4)()
    lib.check(lib.L.orbx_debug_live_resources(a))
    return tuple(a)


def exercise(lib, big):
    import test_bow_frames_batch as t_bow
    import test_emu_mappoint as t_mp
    import test_emu_search as t_search
    import test_keyframe_batch as t_kf
    import test_lastframe_batch as t_lf
    import test_local_points_batch as t_lp
    w, h, nf = (752, 480, 1200) if big else (376, 240, 500)
    t_bow._run(lib, w, h, nf, 3)                         # extractor + vocabulary + resident key frames + FeatureVectors on the device
    t_kf._run(lib, w, h, nf, 3)
    if big:
        t_lf._run(lib, w, h, nf, 3, False)
        t_lp._run(lib, w, h, nf, 3, 900, False)
    t_lp._run(lib, w, h, nf, 3, 900, True)               # resident map points, depth images
    t_mp.run(lib, 40, 60, 1)
    t_search.run_all(lib, w, h, nf, 300, [0])
    # handles used directly: several in flight, the graph, the undistortion model, caller-owned buffers, a communicator
    exs = [ORBextractor(nf, 1.2, 8, 20, 7, lib=lib) for _ in range(3)]
    imgs = np.stack([synth.corner_field(w, h, seed=s, nrect=int(3000 * w * h / (752 * 480))) for s in range(4)])
    for i, ex in enumerate(exs):
        lib.check(lib.L.orbx_set_graph_replay(ex._h, i & 1))
        ex.extract_batch(imgs[:2 + i % 2])
        ex.extract_batch(imgs)                           # buffers grow
        lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 2, 2, 40.0, 0.1))
        M.StereoFetch(ex, 2)
        M.StereoFishEyeKnn(ex, ex, 0, 2, 2)
    # the remaining buffer families of a handle: input pre-step (maps, taps, intermediate frame), undistorted keypoints, the all-pairs distance
    # matrix, the quadtree's phase probe
    ex = exs[0]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    ex.set_input(3, True, 0, remap=(xx + 0.25, yy - 0.5))
    ex.extract_batch(np.repeat(imgs[:2, :, :, None], 3, axis=3))
    ex.set_input(1, True, 0, resize=(w, h))
    ex.extract_batch(np.repeat(np.repeat(imgs[:2], 2, axis=1), 2, axis=2))
    ex.set_input(None)
    ex.set_undistort((300.0, 300.0, w / 2, h / 2), (0.1, -0.05, 1e-3, 1e-3, 0.01))
    ex.profile(True, serial=True)
    r = ex.extract_batch(imgs[:2])
    ex.fetch_undistorted()
    prof = (C.c_longlong * 16)()
    lib.L.orbx_debug_quadtree_profile(ex._h, prof)
    ex.profile(False)
    ex.set_undistort(None)
    M.ORBmatcher.DescriptorDistance(ex, r[0][2][:50], r[1][2][:70])
    dp, hp = C.c_void_p(), C.c_void_p()
    lib.check(lib.L.orbx_device_alloc(exs[0]._h, 1 << 20, C.byref(dp)))
    lib.check(lib.L.orbx_host_alloc(exs[0]._h, 1 << 20, C.byref(hp)))
    comm = multi.Communicator(lib, 1, 0, multi.Communicator.unique_id(lib), device_id=0)
    comm.all_gather(exs[0]); comm.wait(); comm.fetch()
    held = live(lib)
    comm.close()
    lib.check(lib.L.orbx_device_free(exs[0]._h, dp)); lib.check(lib.L.orbx_host_free(exs[0]._h, hp))
    for ex in exs: ex.close()
    return held


def check(lib, big):
    before = live(lib)
    held = exercise(lib, big)
    assert held[0] > before[0] + 20 and held[1] > before[1] and held[2] >= before[2] + 9 and held[3] > before[3] + 60, (before, held)   # the counters do count
    after = live(lib)
    assert after == before, "device buffers / page-locked buffers / streams / events held: %s before, %s after" % (before, after)
    if big:                                               # (the emulator takes a minute per cycle; the facade test below cycles three times)
        exercise(lib, big)
        assert live(lib) == before


def test_everything_is_given_back_emulator(emu_lib):
    check(emu_lib, False)


def test_failed_calls_hold_nothing(emu_lib):
    lib = emu_lib
    before = live(lib)
    h = C.c_void_p()
    assert lib.L.orbx_create(C.byref(h), 0, 1.2, 8, 20, 7, 0) != 0          # refused arguments
    assert lib.L.orbx_create(C.byref(h), 500, 1.2, 8, 20, 7, 99) != 0       # no such device
    ex = ORBextractor(500, 1.2, 8, 20, 7, lib=lib)
    mid = live(lib)
    with pytest.raises(Exception):
        ex.extract_batch(np.zeros((1, 8, 8), np.uint8))                               # an image smaller than the pyramid allows
    kf = C.c_void_p()
    assert lib.L.orbm_keyframe_create(ex._h, None, C.byref(kf)) != 0
    assert lib.L.orbm_points_create(ex._h, None, C.byref(kf)) != 0
    ex.close()
    assert live(lib) == before and mid != before


@pytest.mark.gpu
def test_everything_is_given_back_gpu(hip_lib):
    """+ the device's own account of free memory (hipMemGetInfo of the runtime the library is bound to).  PyTorch is imported in between on purpose:
    mapped after the library it brings a second HIP runtime and its own RCCL into the process, and the communicator of `exercise` has to keep
    working on the library's (tests/test_multi_comm.py)."""
    lib = hip_lib
    runtimes = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))
    assert len(runtimes) == 1, runtimes                   # the library's (nothing else of this test session has brought one)
    hip = C.CDLL(runtimes[0])
    import torch  # noqa: F401

    def free_bytes():
        gc.collect()
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value
    exercise(lib, True)                                   # first use: code objects, the runtime's own pools
    free0 = free_bytes()
    check(lib, True)
    free1 = free_bytes()
    assert free1 >= free0 - (64 << 20), "device memory: %d MB free before, %d MB after" % (free0 >> 20, free1 >> 20)

def test_every_buffer_member_is_released_in_destroy():
    """Source audit: the handle structs hold plain DevBuf / HostBuf members (no destructors); each of them has to appear in the destroy function."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "orb_slam3_detailed_comments_amd", "csrc")
    rd = lambda n: open(os.path.join(root, n)).read()

    def members(src, struct):
        body = src[src.index("struct %s {" % struct):]
        body = body[:body.index("\n};")]
        out = []
        for m in re.finditer(r"(?:orbx::)?(?:DevBuf|HostBuf)<[^>]+>\s+([^;]+);", body):
            out += [re.sub(r"\[.*", "", n.strip()) for n in m.group(1).split(",")]
        return out

    def destroy_body(src, signature):
        d = src[src.index(signature):]
        return d[:d.index("\n}\n")]

    cases = [("orbx_internal.h", "orbx_extractor", "orbx_api.cpp", "void orbx_destroy(orbx_extractor* h)", "h"),
             ("orbv_api.cpp", "orbv_vocabulary", "orbv_api.cpp", "void orbv_destroy(orbv_vocabulary* v)", "v"),
             ("orbx_comm.cpp", "orbx_comm", "orbx_comm.cpp", "void orbx_comm_destroy(orbx_comm* c)", "c")]
    total = 0
    for hdr, struct, cpp, sig, var in cases:
        names = members(rd(hdr), struct)
        body = destroy_body(rd(cpp), sig)
        for n in names:
            if n == "d_status": continue                  # an alias into d_nm (orbx_api.cpp: "counts | mono indices | status word: one block")
            looped = re.search(r"for \(auto& x : %s->%s\) x\.release\(\)" % (var, n), body)
            assert looped or re.search(r"%s->%s\.release\(\)" % (var, n), body), "%s::%s is never released in %s" % (struct, n, sig)
        total += len(names)
    assert total > 60


def test_facade_worlds_leave_nothing_behind(tmp_path, emu_lib):
    """The header-only facade (include/orb_slam3_amd/ORBmatcher.h) over the emulator library: worlds of key frames, frames and map points are
    built, searched through all thirteen methods (the implicit resident cache included) and destroyed twice in one process (three times under
    AddressSanitizer: profiles/r05_final/sanitizers.txt)."""
    import os
    import subprocess
    import sys
    import oracle_lib as ol
    facade = os.path.join(ol.ROOT, "oracle", "_ref", "libmw_facade.so")
    if not os.path.exists(facade):
        pytest.skip("oracle/_ref/libmw_facade.so not built (needs /root/reference)")
    orbx = os.path.join(ol.ROOT, "tests", "emu", "liborbx_emu.so")
    for seed, variant in [(1, "base"), (6, "kb8")]:
        dst = str(tmp_path / ("w_%s.npz" % variant))
        r = subprocess.run([sys.executable, os.path.join(ol.ROOT, "tests", "matcher_world.py"), facade, orbx, str(seed), variant, dst, "twice"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        z = np.load(dst)
        assert z["live_first"][2] >= 3 and np.array_equal(z["live_first"], z["live_again"]), (variant, z["live_first"], z["live_again"])


def test_many_host_threads_with_their_own_handles(emu_lib):
    """Eight host threads, each with its own extractor handle, extract, associate and search at the same time (the documented threading model: a handle belongs
    to one thread at a time; the library's shared state - error strings, the runtime's counters, the emulator's worker pool - has to cope); every thread gets
    exactly what a serial run gets, and nothing stays behind."""
    import threading
    import search_scenes as sc
    lib = emu_lib
    before = live(lib)
    NT, ITER = 8, 3
    pairs = [synth.stereo_pair(376, 240, seed=40 + i, nrect=800) for i in range(NT)]

    def one(ex, i, fv=None, mps=None):
        r = ex.extract_batch(np.stack(pairs[i])); u, d, n = M.ComputeStereoMatches(ex, ex, 47.9, 0.11, 0, 1, 1)
        if fv is None:
            rng = np.random.default_rng(i); fv, k, dd, uu, scales = sc.frame_from_image(pairs[i][0], 500, rng)
            mps = sc.map_points_for_frame(k, dd, uu, scales, 300, rng, 376, 240)
        a = M.ORBmatcher(0.8).SearchByProjection(ex, fv, mps, 3.0, False, 0.0)
        return (r[0][1].tobytes(), r[0][2].tobytes(), u.tobytes(), int(n[0]), a[0], a[1].tobytes()), fv, mps
    exp = []
    for i in range(NT):
        ex = ORBextractor(500, 1.2, 8, 20, 7, lib=lib); exp.append(one(ex, i)); ex.close()
    errs = []

    def work(i):
        try:
            ex = ORBextractor(500, 1.2, 8, 20, 7, lib=lib)
            for it in range(ITER):
                if one(ex, i, exp[i][1], exp[i][2])[0] != exp[i][0]: errs.append((i, it))
            ex.close()
        except Exception as e:                          # noqa: BLE001
            errs.append((i, repr(e)[:200]))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(NT)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert errs == [] and exp[0][0][3] > 50 and exp[0][0][4] > 20
    del exp
    assert live(lib) == before
