"""C1 on the device: Frame::isInFrustum (src/Frame.cc:667-773) + Pinhole::project (src/CameraModels/Pinhole.cpp:61-68) + MapPoint::PredictScale
(src/MapPoint.cc:688-731) as k_frustum, alone (orbm_is_in_frustum) and in front of ORBmatcher::SearchByProjection(Frame, MapPoints)
(orbm_search_local_points = the device part of Tracking::SearchLocalPoints, src/Tracking.cc:4009-4067).

Checker: the reference's own Frame.cc (SetPose, isInFrustum) and ORBmatcher.cc (SearchByProjection) compiled in place (oracle/_ref/libref_frame.so)
on the reference's own stereo Frame.  Bar: every tracking field bit-identical (fp32 operation order reproduced), the same keypoint -> map point
assignment."""
import numpy as np
import pytest

import oracle_lib as ol
from orb_slam3_detailed_comments_amd import ORBextractor, ComputeStereoMatches, synth, views
from orb_slam3_detailed_comments_amd import matcher as M

pytestmark = pytest.mark.skipif(ol.reference_frame_lib() is None, reason="oracle/_ref/libref_frame.so not built (needs /root/reference)")
FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375
BF = FX * 0.110074


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)


def _scene(F, rng, R, t, M):
    """M map points: most of them back-projections of the frame's keypoints (with their descriptors, a few bits flipped), the rest anywhere."""
    N = F.N
    src = rng.integers(0, N, M)
    z = rng.uniform(0.6, 18.0, M)
    u = F.keys["x"][src] + rng.normal(0, 1.2, M); v = F.keys["y"][src] + rng.normal(0, 1.2, M)
    Xc = np.stack([(u - CX) / FX * z, (v - CY) / FY * z, z], 1)
    far = rng.uniform(size=M) < 0.25
    Xc[far] = np.stack([rng.uniform(-12, 12, far.sum()), rng.uniform(-8, 8, far.sum()), rng.uniform(-3, 25, far.sum())], 1)
    Rw = R.astype(np.float64); Xw = (Rw.T @ (Xc - t.astype(np.float64)).T).T
    Ow = -(Rw.T @ t.astype(np.float64))
    d = np.linalg.norm(Xw - Ow, axis=1)
    normal = (Xw - Ow) / d[:, None]
    tilt = rng.normal(0, 0.6, (M, 3)); normal = normal + tilt * (rng.uniform(size=(M, 1)) < 0.5); normal /= np.linalg.norm(normal, axis=1)[:, None]
    lvl = F.keys["octave"][src].astype(np.float64)
    maxd = d * 1.2 ** lvl * rng.uniform(0.7, 1.6, M); mind = maxd / 1.2 ** 7
    desc = F.desc[src].copy()
    for i in range(M):
        for b in rng.choice(256, int(rng.integers(0, 40)), replace=False):
            desc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return (Xw.astype(np.float32), normal.astype(np.float32), mind.astype(np.float32), maxd.astype(np.float32), rng.uniform(size=M) < 0.04, rng.uniform(size=M) < 0.9, desc)


def _check(lib, seeds, npts=5000):
    for seed in seeds:
        rng = np.random.default_rng(seed)
        w, h, nf = [(752, 480, 1200), (640, 480, 1000)][seed & 1]
        L, R = synth.stereo_pair(w, h, seed=seed)
        F = ol.ReferenceFrame(L, R, nf, fx=FX, fy=FY, cx=CX, cy=CY, bf=BF)
        ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
        (_, kL, dL), _ = ex.extract_batch(np.stack([L, R]))
        u, dep, _ = ComputeStereoMatches(ex, ex, BF, F.mb, 0, 1, 1)
        assert kL.tobytes() == F.keys.tobytes() and u[0, :F.N].tobytes() == F.u_right.tobytes()
        sfs = ex.GetScaleFactors()
        fv = views.frame_view(kL, dL, sfs, w, h, u_right=u[0, :F.N], mbf=BF)
        Rcw = _rot(0.02, -0.03, 0.01); tcw = np.array([0.3, -0.1, 0.25], np.float32)
        pos, normal, mind, maxd, bad, obs, desc = _scene(F, rng, Rcw, tcw, npts)
        rp = M.ResidentPoints(ex, pos, normal, mind, maxd, desc)
        for th, far in ((1.0, False), (3.0, True)):
            ref_tr, ref_as, ref_n = F.search_local_points(Rcw, tcw, pos, normal, mind, maxd, bad, obs, desc, 0.5, True, th, far, 9.0, 0.8)
            tr, asg, n = M.SearchLocalPoints(ex, fv, Rcw, tcw, (FX, FY, CX, CY), (0.0, float(w), 0.0, float(h)), BF, sfs, pos, normal, mind, maxd, bad, obs, desc,
                                             0.5, th, far, 9.0, 0.8)
            inv = ref_tr["in_view"]
            assert 0.2 * npts < inv.sum() < 0.9 * npts
            assert np.array_equal(tr["in_view"].astype(bool), inv), "mbTrackInView"
            # mTrackProjX / Y are also written for points that fail the later tests; the others only for points in view
            for k in ("proj_x", "proj_y"):
                assert tr[k].tobytes() == ref_tr[k].tobytes(), k
            for k in ("proj_xr", "depth", "view_cos"):
                assert tr[k][inv].tobytes() == ref_tr[k][inv].tobytes(), k
            assert np.array_equal(tr["scale_level"][inv], ref_tr["scale_level"][inv]), "mnTrackScaleLevel"
            assert n == ref_n and ref_n > npts // 10 and np.array_equal(asg, ref_as), "SearchByProjection assignment differs (%d vs %d matches)" % (n, ref_n)
            # the same with the points resident on the device (orbm_points): only the frame and the flags travel
            tr3, asg3, n3 = M.SearchLocalPoints(ex, fv, Rcw, tcw, (FX, FY, CX, CY), (0.0, float(w), 0.0, float(h)), BF, sfs, pos, normal, mind, maxd, bad, obs, desc,
                                                0.5, th, far, 9.0, 0.8, resident=rp)
            assert n3 == n and np.array_equal(asg3, asg) and all(tr3[k].tobytes() == tr[k].tobytes() for k in tr), "resident points"
            # the producer alone
            tr2, _, _ = M.SearchLocalPoints(ex, fv, Rcw, tcw, (FX, FY, CX, CY), (0.0, float(w), 0.0, float(h)), BF, sfs, pos, normal, mind, maxd, search=False)
            assert np.array_equal(tr2["in_view"], tr["in_view"]) and tr2["proj_x"].tobytes() == tr["proj_x"].tobytes()
        rp.close(); ex.close()


def test_local_points_emulated(emu_lib):
    _check(emu_lib, (3,), npts=1500)


@pytest.mark.gpu
def test_local_points_gpu(hip_lib):
    _check(hip_lib, (3, 4, 5, 6))
