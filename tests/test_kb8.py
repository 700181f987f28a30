"""Kannala-Brandt (fisheye) branches: the triangulation gate of Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1564-1584 ->
KannalaBrandt8::TriangulateMatches, src/CameraModels/KannalaBrandt8.cpp:439-523) and the fisheye branch of ORBmatcher::SearchForTriangulation
(src/ORBmatcher.cc:1203-1240 -> KannalaBrandt8::epipolarConstrain, :322-328).

The checker is the reference's own code throughout: Frame.cc / ORBmatcher.cc AND src/CameraModels/KannalaBrandt8.cpp + Pinhole.cpp compiled
unmodified and in place (oracle/Makefile, CAMSRC) over the stand-in Eigen of oracle/slam_shim (eigen_small.h: Matrix<float,3,4>, comma
initialisers, JacobiSVD<Matrix4f> restated from Eigen 3.3.7 in fp32 - Eigen is an absent external dependency).  The device follows the same
statements with bit-for-bit models of glibc's atan2f / tanf / cosf / sinf and the same fp32 two-sided Jacobi SVD (csrc/kb8_model.h).
Bar: identical accept sets / match pairs, mvDepth and mvStereo3Dpoints identical to the bit."""
import numpy as np
import pytest

import oracle_lib as ol
from orb_slam3_detailed_comments_amd import ORBextractor, synth, sophus
from orb_slam3_detailed_comments_amd import matcher as M

pytestmark = pytest.mark.skipif(ol.reference_frame_lib() is None, reason="oracle/_ref/libref_frame.so not built (needs /root/reference)")

# Examples/Stereo/TUM-VI.yaml:11-32 (TUM-VI room, 512 x 512): the two Kannala-Brandt cameras and the left-to-right transform
CAM1 = [190.978477, 190.973307, 254.931706, 256.897442, 0.003482389402, 0.000715034845, -0.002053236141, 0.000202936736]
CAM2 = [190.442369, 190.434438, 252.598711, 254.917238, 0.003400603976, 0.001766924711, -0.002663898171, 0.000329921072]
RLR = np.array([[0.999999445773493, 0.000791687752817, 0.000694034010224],
                [-0.000823363992158, 0.998899461915674, 0.046895490788700],
                [-0.000656143613422, -0.046896036240590, 0.998899559977407]], np.float32)
TLR = np.array([0.100931237881590, 0.000570764538347, 0.001046438762054], np.float32)      # |t| ~ 0.101 m
# what the frame holds: mTlr = SE3f(Rlr, tlr) keeps a unit quaternion, mRlr = mTlr.rotationMatrix() (src/Frame.cc:1498-1501) - not RLR's bits
MRLR = sophus.SE3f(RLR, TLR).rotationMatrix()


def _fisheye_pair(seed, w=512, h=512):
    """A pair whose disparities are what the rig would see: the right image is the left one shifted by a per-band disparity (1..24 px, depths
    of 0.8 m and more at fx = 191 px, b = 0.1 m).  Near the principal point most pairs triangulate well; off axis and at band seams the
    reprojection test rejects - both outcomes of the gate occur."""
    return synth.stereo_pair(w, h, seed=seed, nrect=2000, max_disp=24, band=64)


def _check(lib, seeds, lap, nf):
    total = 0
    for seed in seeds:
        L, R = _fisheye_pair(seed)
        F = ol.reference_fisheye_frame(L, R, lap, lap, nf, cams=(CAM1, CAM2, RLR, TLR))
        ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
        (mL, kL, dL), (mR, kR, dR) = ex.extract_batch(np.stack([L, R]), lap)
        assert kL.tobytes() == F["keys"].tobytes() and kR.tobytes() == F["keys_right"].tobytes()
        out = M.ComputeStereoFishEyeMatches(ex, ex, CAM1, CAM2, MRLR, TLR, 0, 1, 1)
        nl, nr = len(kL), len(kR)
        acc_ref = F["l2r"] >= 0
        assert acc_ref.sum() > 15 and (F["l2r"] < 0).sum() > 15
        assert np.array_equal(out["l2r"][0, :nl], F["l2r"]), "mvLeftToRightMatch differs from the reference Frame (seed %d)" % seed
        assert np.array_equal(out["r2l"][0, :nr], F["r2l"]), "mvRightToLeftMatch differs from the reference Frame (seed %d)" % seed
        assert out["n"][0] == int(acc_ref.sum())
        d = out["depth"][0, :nl]
        assert np.all(d[~acc_ref] == -1.0)
        assert d[acc_ref].tobytes() == F["depth"][acc_ref].tobytes(), "mvDepth differs from the reference Frame (seed %d): max relative difference %.2e" % (
            seed, (np.abs(d[acc_ref] - F["depth"][acc_ref]) / F["depth"][acc_ref]).max())
        assert out["p3d"][0, :nl][acc_ref].tobytes() == F["p3d"][acc_ref].tobytes(), "mvStereo3Dpoints differs from the reference Frame (seed %d)" % seed
        total += int(acc_ref.sum())
        ex.close()
    return total


def test_fisheye_stereo_gate_emulated(emu_lib):
    assert _check(emu_lib, (5,), (0, 511), 1000) > 15


@pytest.mark.gpu
def test_fisheye_stereo_gate_gpu(hip_lib):
    assert _check(hip_lib, (5, 6, 7, 8), (0, 511), 1500) > 100
    assert _check(hip_lib, (9,), (100, 400), 1000) > 10


# ---- an INDEPENDENT check of the depths: the same triangulation in float64 with numpy's SVD -------------------------------------------------------------
# The bit test above compares the device with the reference's KannalaBrandt8.cpp compiled over a RESTATED Eigen (oracle/slam_shim/eigen_small.h) - two
# renderings of one reading of Eigen's fp32 Jacobi SVD.  This one shares nothing with either: Kannala-Brandt unprojection by Newton's method,
# the DLT system of GeometricTools::Triangulate and numpy.linalg.svd, all in float64.  It cannot be bit-exact (the reference computes in fp32); it bounds how
# far the fp32 chain - whoever restated it - is from the mathematical solution.
def _unproject64(cam, pts):
    fx, fy, cx, cy, k1, k2, k3, k4 = [np.float64(v) for v in cam]
    x = (pts[:, 0].astype(np.float64) - cx) / fx; y = (pts[:, 1].astype(np.float64) - cy) / fy
    td = np.minimum(np.maximum(np.sqrt(x * x + y * y), -np.pi / 2), np.pi / 2)
    th = td.copy()
    for _ in range(20):
        t2 = th * th; t4 = t2 * t2; t6 = t4 * t2; t8 = t4 * t4
        th = th - (th * (1 + k1 * t2 + k2 * t4 + k3 * t6 + k4 * t8) - td) / (1 + 3 * k1 * t2 + 5 * k2 * t4 + 7 * k3 * t6 + 9 * k4 * t8)
    scale = np.where(td > 1e-8, np.tan(th) / np.where(td > 1e-8, td, 1.0), 1.0)
    return np.stack([x * scale, y * scale, np.ones_like(x)], 1)


def _depths64(kL, kR, l2r):
    R12 = MRLR.astype(np.float64); t12 = TLR.astype(np.float64)
    R21 = R12.T; T2 = np.concatenate([R21, (-R21 @ t12)[:, None]], 1); T1 = np.concatenate([np.eye(3), np.zeros((3, 1))], 1)
    idx = np.nonzero(l2r >= 0)[0]
    r1 = _unproject64(CAM1, np.stack([kL["x"][idx], kL["y"][idx]], 1)); r2 = _unproject64(CAM2, np.stack([kR["x"][l2r[idx]], kR["y"][l2r[idx]]], 1))
    z = np.zeros(len(idx)); p = np.zeros((len(idx), 3))
    for i in range(len(idx)):
        A = np.stack([r1[i, 0] * T1[2] - T1[0], r1[i, 1] * T1[2] - T1[1], r2[i, 0] * T2[2] - T2[0], r2[i, 1] * T2[2] - T2[1]])
        v = np.linalg.svd(A)[2][3]
        p[i] = v[:3] / v[3]; z[i] = p[i, 2]
    return idx, z, p


def _depth_accuracy(lib, seeds, nf):
    rel = []
    for seed in seeds:
        L, R = _fisheye_pair(seed)
        ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
        (mL, kL, dL), (mR, kR, dR) = ex.extract_batch(np.stack([L, R]), (0, 511))
        out = M.ComputeStereoFishEyeMatches(ex, ex, CAM1, CAM2, MRLR, TLR, 0, 1, 1)
        l2r = out["l2r"][0, :len(kL)]
        idx, z, p = _depths64(kL, kR, l2r)
        d = out["depth"][0, :len(kL)][idx].astype(np.float64)
        assert np.all(d > 0) and np.all(z > 0)
        rel.append(np.abs(d - z) / z)
        p3 = out["p3d"][0, :len(kL)][idx].astype(np.float64)
        assert np.all(np.linalg.norm(p3 - p, axis=1) <= 2e-2 * np.linalg.norm(p, axis=1) + 1e-6)
        ex.close()
    rel = np.concatenate(rel)
    assert len(rel) > 15
    # fp32 unprojection + a 4x4 fp32 Jacobi SVD at parallax angles above 1.15 degrees (cos < 0.9998): most depths agree to 1e-5, the worst stays within 2 %
    assert np.median(rel) < 1e-4 and np.percentile(rel, 99) < 5e-3 and rel.max() < 2e-2, (np.median(rel), np.percentile(rel, 99), rel.max())
    return len(rel), float(np.median(rel)), float(rel.max())


def test_fisheye_depths_close_to_float64_solution_emulated(emu_lib):
    n, med, worst = _depth_accuracy(emu_lib, (5,), 1000)
    print("fisheye depths vs float64 DLT: %d matches, median relative difference %.2e, worst %.2e" % (n, med, worst))


@pytest.mark.gpu
def test_fisheye_depths_close_to_float64_solution_gpu(hip_lib):
    n, med, worst = _depth_accuracy(hip_lib, (5, 6, 7, 8, 9, 10), 1500)
    print("fisheye depths vs float64 DLT: %d matches, median relative difference %.2e, worst %.2e" % (n, med, worst))
