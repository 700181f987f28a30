"""ORBmatcher drop-in check against the REFERENCE's own src/ORBmatcher.cc.

oracle/_ref/libmw_ref.so is the reference's ORBmatcher.cc + ORBmatcher.h compiled unmodified and in place against the stand-in world of
oracle/slam_shim (Frame / KeyFrame / MapPoint / cameras / poses); oracle/_ref/libmw_facade.so drives include/orb_slam3_amd/ORBmatcher.h —
every public method of the reference class, on top of the C ABI — with the very same objects.  tests/matcher_world.py builds seeded
worlds (900 scene points, noisy descriptors around TH_LOW / TH_HIGH, clutter, bad / unobserved / duplicated map points, vocabulary nodes,
stereo and monocular features, three poses, Sim3 with scale != 1, a fisheye rig) and runs all thirteen methods; every return value
and everything a method wrote (Frame::mvpMapPoints, vpMatched / vpMatchedKF, vnMatches12 + vbPrevMatched, vMatchedPairs, vpReplacePoint, the
key frame's map points and the Replace / AddObservation log of Fuse) must be IDENTICAL.  This pins M1-M6 and the f-rows of DESIGN.md §1.

Not GPU-marked: facade over the CPU emulator build of the kernels.  GPU-marked: facade over the HIP library (C ABI on the device)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
from orb_slam3_detailed_comments_amd import _lib

ROOT = ol.ROOT
REF = os.path.join(ROOT, "oracle", "_ref", "libmw_ref.so")
FACADE = os.path.join(ROOT, "oracle", "_ref", "libmw_facade.so")
# the same driver over the reference's OWN Frame, KeyFrame and MapPoint classes (Frame.cc / KeyFrame.cc / MapPoint.cc linked in, and in the
# facade build compiled against the drop-in ORBmatcher.h); only Map, KeyFrameDatabase, the IMU types, the camera and the Eigen / Sophus algebra
# are stand-ins there
REF_REAL = os.path.join(ROOT, "oracle", "_ref", "libmw_ref_full.so")
FACADE_REAL = os.path.join(ROOT, "oracle", "_ref", "libmw_facade_full.so")
RUNNER = os.path.join(ROOT, "tests", "matcher_world.py")
pytestmark = pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(FACADE)), reason="oracle/_ref/libmw_*.so not built (needs /root/reference)")

MIN_MATCHES = {"sbp_mappoints_0": 100, "sbp_frame_fwd_7": 100, "sbp_keyframe_100": 100, "sbp_sim3_100_0": 100, "bow_frame_1": 80, "bow_keyframes_1": 60,
               "init_0": 30, "triang_0_0": 10, "sim3_100": 60, "fuse": 100, "fuse_th1": 15, "fuse_sim3": 60, "rig_sbp_mappoints_1": 200, "rig_sbp_frame_fwd_1": 300, "rig_fuse_0": 50, "rig_fuse_1": 25, "rig_bow_frame_1": 80, "rig_bow_keyframes_1": 40,
               "kb8_triangulation_rig0_001": 40, "kb8_triangulation_rig1_001": 60, "kb8_triangulation_rig1_011": 60}


def _run(tmp_path, driver, orbx, seed, variant, tag):
    dst = str(tmp_path / ("%s_%d_%s.npz" % (tag, seed, variant)))
    r = subprocess.run([sys.executable, RUNNER, driver, orbx, str(seed), variant, dst], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.load(dst)


def _compare(tmp_path, orbx, cases, ref=REF, facade=FACADE):
    for seed, variant in cases:
        a = _run(tmp_path, ref, "", seed, variant, "ref")
        b = _run(tmp_path, facade, orbx, seed, variant, "facade")
        assert bytes(a["flavour"]) == b"reference" and bytes(b["flavour"]) == b"facade"
        assert set(a.files) == set(b.files) and len(a.files) > 5
        for k in a.files:
            if k == "flavour":
                continue
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), "%s differs from the reference's ORBmatcher.cc (seed %d, %s): %d of %d values, return %d vs %d" % (
                k, seed, variant, int((a[k] != b[k]).sum()), a[k].size, a[k][0], b[k][0])
            if k in MIN_MATCHES:
                assert a[k][0] >= MIN_MATCHES[k], "scene too easy to be a test: %s returned %d" % (k, a[k][0])


def test_matcher_facade_equals_reference_emulated(tmp_path, emu_lib):
    _compare(tmp_path, os.path.join(ROOT, "tests", "emu", "liborbx_emu.so"), [(1, "base"), (2, "dense"), (3, "hard"), (4, "rig"), (5, "rig"), (6, "kb8"), (7, "kb8")])


@pytest.mark.gpu
def test_matcher_facade_equals_reference_gpu(tmp_path, hip_lib):
    _compare(tmp_path, _lib.HIP_LIB_PATH, [(1, "base"), (2, "dense"), (3, "hard"), (4, "rig"), (11, "base"), (12, "dense"), (13, "rig"), (6, "kb8"), (14, "kb8"), (15, "kb8")])


real = pytest.mark.skipif(not (os.path.exists(REF_REAL) and os.path.exists(FACADE_REAL)), reason="oracle/_ref/libmw_*_full.so not built (needs /root/reference)")


@real
def test_real_classes_agree_with_standins(tmp_path):
    """The reference's ORBmatcher.cc gives the same search results over its own Frame / KeyFrame / MapPoint classes as over the stand-ins of
    oracle/slam_shim (Fuse excluded: the real Replace / AddObservation change more state than the stand-in log records)."""
    a = _run(tmp_path, REF_REAL, "", 1, "base", "real")
    b = _run(tmp_path, REF, "", 1, "base", "standin")
    for k in a.files:
        if k != "flavour" and not k.startswith("fuse"):
            assert np.array_equal(a[k], b[k]), k


@real
def test_matcher_facade_equals_reference_real_classes_emulated(tmp_path, emu_lib):
    _compare(tmp_path, os.path.join(ROOT, "tests", "emu", "liborbx_emu.so"), [(1, "base"), (3, "hard"), (4, "rig"), (6, "kb8")], REF_REAL, FACADE_REAL)


@real
@pytest.mark.gpu
def test_matcher_facade_equals_reference_real_classes_gpu(tmp_path, hip_lib):
    _compare(tmp_path, _lib.HIP_LIB_PATH, [(1, "base"), (2, "dense"), (3, "hard"), (4, "rig"), (21, "base"), (22, "rig"), (23, "kb8")], REF_REAL, FACADE_REAL)


def test_matcher_facade_equals_reference_random_parameters_emulated(tmp_path, emu_lib):
    """the same worlds with the methods' parameters (th, nnratio, ORBdist, window, ratioHamming) drawn at random per seed instead of the reference's call-site
    values (variant "fuzz"; tools/soak_world_fuzz.py runs more seeds): Fuse below th = 2.8 is what the fixed values had hidden (round 5)"""
    orbx = os.path.join(ROOT, "tests", "emu", "liborbx_emu.so")
    for seed in (11, 12, 13):
        a = _run(tmp_path, REF, "", seed, "fuzz", "ref")
        b = _run(tmp_path, FACADE, orbx, seed, "fuzz", "facade")
        assert set(a.files) == set(b.files) and len(a.files) > 20
        for k in a.files:
            if k != "flavour":
                assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), "%s differs from the reference's ORBmatcher.cc (seed %d, random parameters)" % (k, seed)
