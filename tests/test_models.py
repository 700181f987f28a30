"""Host checks of the two bit-exact models the device code relies on:
  csrc/libstdcxx_sort_model.h   == libstdc++ std::sort (tie permutation of the quadtree's final rounds)
  csrc/glibc_sincosf_model.h    == glibc cosf/sinf on [0, 2*pi] (BRIEF steering)"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import oracle_lib as ol

ROOT = ol.ROOT
CSRC = os.path.join(ROOT, "orb_slam3_detailed_comments_amd", "csrc")


def test_libstdcxx_sort_model_matches_std_sort():
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "t")
        subprocess.run(["g++", "-O2", "-std=c++14", "-I" + CSRC, os.path.join(ROOT, "tests", "cpp", "sort_model_test.cpp"), "-o", exe], check=True)
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "bad 0" in r.stdout


def test_sincosf_model_matches_glibc_sample():
    src = r'''
    #include "glibc_sincosf_model.h"
    extern "C" float m_cosf(float x) { return orbx::glibc_cosf(x); }
    extern "C" float m_sinf(float x) { return orbx::glibc_sinf(x); }
    '''
    with tempfile.TemporaryDirectory() as td:
        cpp = os.path.join(td, "m.cpp"); so = os.path.join(td, "m.so")
        open(cpp, "w").write(src)
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I" + CSRC, cpp, "-o", so], check=True)
        M = C.CDLL(so)
        for f in (M.m_cosf, M.m_sinf):
            f.restype = C.c_float; f.argtypes = [C.c_float]
        L = ol.oracle()
        rng = np.random.default_rng(0)
        # every angle the extractor can produce is angle_deg*pi/180 with angle_deg in [0,360]; sample densely,
        # plus the exact multiples of 0.25 degree and random bit patterns in [0, 6.5]
        xs = np.concatenate([
            (np.arange(0, 360.25, 0.25, dtype=np.float32) * np.float32(np.pi / 180.0)).astype(np.float32),
            rng.uniform(0, 6.2832, 60000).astype(np.float32),
            rng.integers(0, np.float32(6.5).view(np.uint32), 60000, dtype=np.uint32).view(np.float32),
            np.array([0.0, 1e-30, 2.0 ** -13, 2.0 ** -12, 0.78539, 0.785398163, 0.7853982, 1.5707963, 3.1415927, 4.712389, 6.2831855], np.float32),
        ])
        bad = 0
        for x in xs.tolist():
            bad += np.float32(M.m_cosf(x)).tobytes() != np.float32(L.orbo_cosf(x)).tobytes()
            bad += np.float32(M.m_sinf(x)).tobytes() != np.float32(L.orbo_sinf(x)).tobytes()
        assert bad == 0


def _predict_scale_float(ratio, lsf, nlevels):
    """MapPoint::PredictScale as the reference's translation unit evaluates it: logf (glibc), float division, ceil(float)"""
    import ctypes as C
    libm = C.CDLL("libm.so.6"); libm.logf.restype = C.c_float; libm.logf.argtypes = [C.c_float]
    out = []
    for r in ratio:
        n = int(np.ceil(np.float32(np.float32(libm.logf(float(r))) / np.float32(lsf))))
        out.append(min(max(n, 0), nlevels - 1))
    return np.array(out, np.int32)


def test_predict_scale_uses_float_log():
    """src/MapPoint.cc:714-731 is compiled `using namespace std`, so log(ratio) on a float is logf: pinned on the reference's OWN MapPoint.cc
    (oracle/_ref/libref_mappoint.so).  At ratio = 1.2f - a map point seen from the distance it was created at, one level up - the float and the
    double formula give different levels (1 vs 2), so the device must use the float one (csrc/glibc_logf_model.h in k_frustum)."""
    import ctypes as C
    L = ol.reference_mappoint_lib()
    if L is None:
        pytest.skip("oracle/_ref/libref_mappoint.so not built")
    rng = np.random.default_rng(5)
    special = np.array([1.2, 5.15978241, 0.578703642], np.float32)
    ratio = np.concatenate([special, np.float32(1.2) ** rng.integers(-2, 9, 4000).astype(np.float32), rng.uniform(0.3, 8.0, 4000).astype(np.float32)]).astype(np.float32)
    lsf = np.float32(np.log(np.float64(np.float32(1.2))))
    dist = np.ones(len(ratio), np.float32); out = np.zeros(len(ratio), np.int32)
    L.ref_mp_predict_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    L.ref_mp_predict_scale(ratio.ctypes.data, dist.ctypes.data, len(ratio), float(lsf), 8, out.ctypes.data)
    assert np.array_equal(out, _predict_scale_float(ratio, lsf, 8))
    dbl = np.clip(np.ceil(np.log(ratio.astype(np.float64)) / np.float64(lsf)).astype(np.int32), 0, 7)
    assert out[0] == 1 and dbl[0] == 2 and (out != dbl).sum() >= 2, "the float and the double formula must differ at the powers of 1.2f"


def _frustum_levels(lib, ratio):
    """mnTrackScaleLevel from k_frustum for points on the optical axis at distance 1 with mfMaxDistance = ratio"""
    from orb_slam3_detailed_comments_amd import ORBextractor
    from orb_slam3_detailed_comments_amd import matcher as M
    ex = ORBextractor(100, 1.2, 8, 20, 7, lib=lib)
    n = len(ratio)
    pos = np.tile(np.array([0, 0, 1], np.float32), (n, 1)); normal = pos.copy()
    tr, _, _ = M.SearchLocalPoints(ex, None, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), (400.0, 400.0, 320.0, 240.0), (0.0, 640.0, 0.0, 480.0), 40.0, ex.GetScaleFactors(),
                                   pos, normal, np.full(n, 1e-3, np.float32), ratio, search=False)
    ex.close()
    assert tr["in_view"].all()
    return tr["scale_level"]


def test_device_predict_scale_matches_glibc_logf(emu_lib):
    rng = np.random.default_rng(6)
    ratio = np.concatenate([np.array([1.2, 5.15978241], np.float32), np.float32(1.2) ** rng.integers(0, 9, 500).astype(np.float32), rng.uniform(0.84, 8.0, 3000).astype(np.float32)]).astype(np.float32)
    lsf = np.float32(np.log(np.float64(np.float32(1.2))))
    assert np.array_equal(_frustum_levels(emu_lib, ratio), _predict_scale_float(ratio, lsf, 8))


@pytest.mark.gpu
def test_device_predict_scale_matches_glibc_logf_gpu(hip_lib):
    rng = np.random.default_rng(7)
    ratio = np.concatenate([np.array([1.2, 5.15978241], np.float32), np.float32(1.2) ** rng.integers(0, 9, 5000).astype(np.float32), rng.uniform(0.84, 8.0, 200000).astype(np.float32)]).astype(np.float32)
    lsf = np.float32(np.log(np.float64(np.float32(1.2))))
    assert np.array_equal(_frustum_levels(hip_lib, ratio), _predict_scale_float(ratio, lsf, 8))
