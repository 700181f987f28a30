"""Parity of the launch shape the headline times.

bench.py runs 128 stereo pairs (256 images) per step on each of four extractor handles in flight, the frames resident in pyramid level 0
(orbx_input_buffer), stage events on.  Above ORBX_QT_WIDE_BATCH = 32 images the library switches kernel forms (csrc/orbx_api.cpp: the quadtree on
256 threads per tree, eight keypoints per wave in k_orient_brief, blur and FAST on two streams, eager event records), so the small-batch parity
tests do not execute the code the headline times.  Here EVERY image of every handle's last batch - keypoints, descriptors, mvuRight, mvDepth - is
compared with the reference's own stereo Frame constructor (src/Frame.cc:105-230, oracle/_ref/libref_frame.so), at B = 64 and B = 256 images,
with hipGraph replay on and off, with the stage events on and off; likewise the fisheye rig (512x512, nFeatures 1500, 64 pairs per handle) and the
batched Tracking::SearchLocalPoints at B = 128 frames."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
import headline_check as hc
from cases import EUROC_BF, EUROC_B
from orb_slam3_detailed_comments_amd import ORBextractor, synth
from orb_slam3_detailed_comments_amd import matcher as M

FX = 458.654


def _in_flight(handles, enqueue, fetch, rounds):
    """bench.py's schedule: a handle's batch is fetched when all handles have one in flight"""
    pending = []
    for s in range(rounds * len(handles)):
        i = s % len(handles)
        if len(pending) == len(handles):
            fetch(pending.pop(0))
        enqueue(i)
        pending.append(i)
    while pending:
        fetch(pending.pop(0))


def _stereo_shape(lib, W, H, nf, P, NH, graph, profile, nunique, nrect=3000, min_matches=100):
    uniq = [synth.stereo_pair(W, H, seed=7000 + i, nrect=nrect) for i in range(nunique)]
    exp = [hc.StereoExpectation(l, r, nf, FX, EUROC_BF, EUROC_B) for l, r in uniq]
    handles = [ORBextractor(nf, 1.2, 8, 20, 7, lib=lib) for _ in range(NH)]
    order = [[(p * (2 * i + 1) + 5 * i) % nunique for p in range(P)] for i in range(NH)]          # every handle sees the pairs in another order
    ups, outs = [], []
    for i, h in enumerate(handles):
        batch = np.stack([uniq[u][0] for u in order[i]] + [uniq[u][1] for u in order[i]])
        ups.append(h.input_upload(batch))
        h.graph_replay(graph); h.profile(profile)
        cap = h.max_keypoints()
        outs.append(dict(k=h.pinned_empty((2 * P, cap, 28), np.uint8), d=h.pinned_empty((2 * P, cap, 32), np.uint8), n=np.zeros(2 * P, np.int32), m=np.zeros(2 * P, np.int32),
                         u=h.pinned_empty((P, cap), np.float32), z=h.pinned_empty((P, cap), np.float32), nm=np.zeros(P, np.int32)))
    cap = handles[0].max_keypoints()

    def enqueue(i):
        h = handles[i]; p, shape, st, ist = ups[i]
        h.enqueue(None, (0, 0), device_ptr=p, shape=shape, stride=st, image_stride=ist)
        lib.check(lib.L.orbm_stereo_match(h._h, 0, h._h, P, P, EUROC_BF, EUROC_B))

    def fetch(i):
        h, o = handles[i], outs[i]
        for a in o.values():
            a[...] = 0
        lib.check(lib.L.orbx_fetch(h._h, o["k"].ctypes.data, o["d"].ctypes.data, cap, o["n"].ctypes.data, o["m"].ctypes.data))
        lib.check(lib.L.orbm_stereo_fetch(h._h, P, o["u"].ctypes.data, o["z"].ctypes.data, cap, o["nm"].ctypes.data))

    _in_flight(handles, enqueue, fetch, 2)
    checked = 0
    for i in range(NH):
        o = outs[i]
        for p in range(P):
            e = exp[order[i][p]]
            bad = e.differences(o["k"][p], o["d"][p], o["n"][p], o["k"][P + p], o["d"][P + p], o["n"][P + p], o["u"][p], o["z"][p], o["nm"][p])
            assert not bad, "handle %d pair %d (scene %d): %s differ from the %s" % (i, p, order[i][p], ", ".join(bad), e.kind)
            assert o["m"][p] == o["n"][p] and e.nm > min_matches                   # lapping {0, 0}: monoIndex = N
            checked += 1
    for h in handles:
        h.close()
    return checked, exp[0].kind


def test_headline_shape_emulated(emu_lib):
    """the same checker on the CPU build of the kernels: 34 images per handle (> ORBX_QT_WIDE_BATCH, the large-batch forms), two handles"""
    n, _ = _stereo_shape(emu_lib, 376, 240, 300, 17, 2, False, True, 3, nrect=800, min_matches=40)
    assert n == 34


@pytest.mark.gpu
@pytest.mark.parametrize("P,graph,profile", [(32, False, True), (32, True, False), (128, False, True), (128, False, False), (128, True, False)],
                         ids=["B64-events", "B64-graph", "B256-events", "B256-plain", "B256-graph"])
def test_headline_shape_stereo_gpu(hip_lib, P, graph, profile):
    n, kind = _stereo_shape(hip_lib, 752, 480, 1200, P, 4, graph, profile, 32)
    assert n == 4 * P
    if ol.reference_frame_lib() is not None:
        assert kind == "reference"


@pytest.mark.gpu
@pytest.mark.skipif(ol.reference_frame_lib() is None, reason="oracle/_ref/libref_frame.so not built (needs /root/reference)")
def test_headline_shape_fisheye_gpu(hip_lib):
    """BASELINE.json configs[2] at the size bench.py --config fisheye runs it: 64 pairs per handle, four handles, lapping {0, 511}"""
    from test_kb8 import CAM1, CAM2, RLR, MRLR, TLR, _fisheye_pair
    lib, P, NH, nf, lap, nunique = hip_lib, 64, 4, 1500, (0, 511), 16
    uniq = [_fisheye_pair(400 + i) for i in range(nunique)]
    handles = [ORBextractor(nf, 1.2, 8, 20, 7, lib=lib) for _ in range(NH)]
    order = [[(p * (2 * i + 1) + 3 * i) % nunique for p in range(P)] for i in range(NH)]

    kb = M.KB8Stereo()
    kb.cam1[:] = CAM1; kb.cam2[:] = CAM2; kb.R12[:] = MRLR.ravel().tolist(); kb.t12[:] = TLR.tolist()
    ups, outs = [], []
    for i, h in enumerate(handles):
        ups.append(h.input_upload(np.stack([uniq[u][0] for u in order[i]] + [uniq[u][1] for u in order[i]])))
        h.profile(True)
        cap = h.max_keypoints()
        outs.append(dict(k=np.zeros((2 * P, cap, 28), np.uint8), d=np.zeros((2 * P, cap, 32), np.uint8), n=np.zeros(2 * P, np.int32), m=np.zeros(2 * P, np.int32),
                         l2r=np.zeros((P, cap), np.int32), r2l=np.zeros((P, cap), np.int32), z=np.zeros((P, cap), np.float32), p3=np.zeros((P, cap, 3), np.float32),
                         nm=np.zeros(P, np.int32)))
    cap = handles[0].max_keypoints()

    def enqueue(i):
        h = handles[i]; p, shape, st, ist = ups[i]
        h.enqueue(None, lap, device_ptr=p, shape=shape, stride=st, image_stride=ist)
        lib.check(lib.L.orbm_stereo_fisheye(h._h, 0, h._h, P, P, C.byref(kb)))

    def fetch(i):
        h, o = handles[i], outs[i]
        lib.check(lib.L.orbx_fetch(h._h, o["k"].ctypes.data, o["d"].ctypes.data, cap, o["n"].ctypes.data, o["m"].ctypes.data))
        lib.check(lib.L.orbm_stereo_fisheye_fetch(h._h, P, o["l2r"].ctypes.data, o["r2l"].ctypes.data, o["z"].ctypes.data, o["p3"].ctypes.data, o["nm"].ctypes.data, cap))

    _in_flight(handles, enqueue, fetch, 2)
    # the reference per scene once; every pair of every handle against it
    cache = {}
    total = 0
    for i in range(NH):
        o = outs[i]
        for p in range(P):
            u = order[i][p]
            key = (u,)
            if key not in cache:
                bad = hc.fisheye_differences(uniq[u][0], uniq[u][1], lap, nf, (CAM1, CAM2, RLR, TLR), o["k"][p], o["d"][p], o["n"][p], o["k"][P + p], o["d"][P + p], o["n"][P + p],
                                             o["l2r"][p], o["r2l"][p], o["z"][p], o["p3"][p], o["nm"][p])
                assert not bad, "handle %d pair %d (scene %d): %s" % (i, p, u, ", ".join(bad))
                cache[key] = (i, p)
            else:       # the same scene elsewhere in a batch / on another handle: byte-identical to the checked instance
                j, q = cache[key]; r = outs[j]
                for a, b, name in ((o["k"][p], r["k"][q], "mvKeys"), (o["k"][P + p], r["k"][P + q], "mvKeysRight"), (o["d"][p], r["d"][q], "mDescriptors"),
                                   (o["d"][P + p], r["d"][P + q], "mDescriptorsRight"), (o["l2r"][p], r["l2r"][q], "l2r"), (o["r2l"][p], r["r2l"][q], "r2l"),
                                   (o["z"][p], r["z"][q], "mvDepth"), (o["p3"][p], r["p3"][q], "mvStereo3Dpoints")):
                    assert a.tobytes() == b.tobytes(), "handle %d pair %d: %s differs from the checked instance of scene %d" % (i, p, name, u)
                assert o["nm"][p] == r["nm"][q]
            total += int(o["nm"][p])
    assert len(cache) == nunique and total > 15 * NH * P
    for h in handles:
        h.close()


@pytest.mark.gpu
@pytest.mark.skipif(ol.reference_frame_lib() is None, reason="oracle/_ref/libref_frame.so not built (needs /root/reference)")
def test_headline_shape_local_points_gpu(hip_lib):
    """BASELINE.json configs[3] at bench.py's size: 128 RGB-D-shaped frames per batch against a resident 5000-point map (the single-frame call
    that checks each frame is pinned to the reference's Frame.cc + ORBmatcher.cc in tests/test_local_points.py)"""
    from test_local_points_batch import _large_batch
    _large_batch(hip_lib, 640, 480, 1000, 128, 5000, 12)
