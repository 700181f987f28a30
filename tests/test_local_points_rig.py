"""C1 for Kannala-Brandt cameras and the two-camera rig: k_frustum with camera_type = 1 and Frame::isInFrustumChecks
(src/Frame.cc:1592-1650, one call per camera: mR = Rrl * mRcw, mt = Rrl * mtcw + trl, twc = mRwc * tlr + mOw, mpCamera2) on the device, feeding
ORBmatcher::SearchByProjection with its right-camera branch (src/ORBmatcher.cc:170-236) - orbm_is_in_frustum_rig / orbm_search_local_points_fisheye.

Checker: the reference's own Frame.cc (fisheye-rig constructor, SetPose, isInFrustum) and ORBmatcher.cc compiled in place
(oracle/_ref/libref_frame.so) over the reference's own KannalaBrandt8.cpp (KannalaBrandt8::project :87-104, compiled unmodified:
DESIGN.md section 2).  Bar: mbTrackInView / mbTrackInViewR and every stored field
bit-identical, the same keypoint -> map point assignment over both cameras."""
import numpy as np
import pytest

import oracle_lib as ol
from orb_slam3_detailed_comments_amd import ORBextractor, synth, views
from orb_slam3_detailed_comments_amd import matcher as M
from test_kb8 import CAM1, CAM2, RLR, TLR, _fisheye_pair
from test_local_points import _rot

pytestmark = pytest.mark.skipif(ol.reference_frame_lib() is None, reason="oracle/_ref/libref_frame.so not built (needs /root/reference)")


def _kb8_unproject(cam, u, v):
    """approximate inverse of the Kannala-Brandt projection (Newton on theta), float64: only used to PLACE map points in front of keypoints"""
    fx, fy, cx, cy, k0, k1, k2, k3 = cam
    x = (u - cx) / fx; y = (v - cy) / fy
    r = np.sqrt(x * x + y * y); th = r.copy()
    for _ in range(12):
        t2 = th * th
        f = th * (1 + t2 * (k0 + t2 * (k1 + t2 * (k2 + t2 * k3)))) - r
        df = 1 + t2 * (3 * k0 + t2 * (5 * k1 + t2 * (7 * k2 + 9 * t2 * k3)))
        th = th - f / df
    s = np.where(r > 1e-9, np.tan(th) / np.maximum(r, 1e-9), 1.0)
    return np.stack([x * s, y * s, np.ones_like(x)], 1)


def _scene(F, rng, Rcw, tcw, npts):
    """map points on the rays of left AND right keypoints (their descriptors, a few bits flipped), some anywhere, some behind / far / tilted"""
    nl, nr = F.nl, F.nr
    Rlr = RLR.astype(np.float64); tlr = TLR.astype(np.float64)
    side = rng.uniform(size=npts) < 0.5
    src = np.where(side, rng.integers(0, nl, npts), rng.integers(0, nr, npts))
    u = np.where(side, F.keys["x"][np.minimum(src, nl - 1)], F.keys_right["x"][np.minimum(src, nr - 1)]) + rng.normal(0, 1.0, npts)
    v = np.where(side, F.keys["y"][np.minimum(src, nl - 1)], F.keys_right["y"][np.minimum(src, nr - 1)]) + rng.normal(0, 1.0, npts)
    z = rng.uniform(0.8, 12.0, npts)
    rayL = _kb8_unproject(CAM1, u, v) * z[:, None]
    rayR = _kb8_unproject(CAM2, u, v) * z[:, None]
    XcL = np.where(side[:, None], rayL, (Rlr @ rayR.T).T + tlr)          # camera-1 coordinates (Tlr maps camera 2 -> camera 1)
    far = rng.uniform(size=npts) < 0.2
    XcL[far] = np.stack([rng.uniform(-10, 10, far.sum()), rng.uniform(-8, 8, far.sum()), rng.uniform(-3, 20, far.sum())], 1)
    Rw = Rcw.astype(np.float64); Xw = (Rw.T @ (XcL - tcw.astype(np.float64)).T).T
    Ow = -(Rw.T @ tcw.astype(np.float64))
    d = np.linalg.norm(Xw - Ow, axis=1)
    normal = (Xw - Ow) / d[:, None]
    tilt = rng.normal(0, 0.6, (npts, 3)); normal = normal + tilt * (rng.uniform(size=(npts, 1)) < 0.5); normal /= np.linalg.norm(normal, axis=1)[:, None]
    octv = np.where(side, F.keys["octave"][np.minimum(src, nl - 1)], F.keys_right["octave"][np.minimum(src, nr - 1)]).astype(np.float64)
    maxd = d * 1.2 ** octv * rng.uniform(0.7, 1.6, npts); mind = maxd / 1.2 ** 7
    desc = np.where(side[:, None], F.desc[np.minimum(src, nl - 1)], F.desc[nl + np.minimum(src, nr - 1)]).copy()
    for i in range(npts):
        for b in rng.choice(256, int(rng.integers(0, 40)), replace=False):
            desc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return (Xw.astype(np.float32), normal.astype(np.float32), mind.astype(np.float32), maxd.astype(np.float32), rng.uniform(size=npts) < 0.04, rng.uniform(size=npts) < 0.9, desc)


def _check(lib, seeds, npts, w=512, h=512, nf=1500, lap=(0, 511)):
    for seed in seeds:
        rng = np.random.default_rng(100 + seed)
        L, R = _fisheye_pair(seed, w, h)
        F = ol.ReferenceRigFrame(L, R, lap, lap, nf, (CAM1, CAM2, RLR, TLR))
        ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
        (mL, kL, dL), (mR, kR, dR) = ex.extract_batch(np.stack([L, R]), lap)
        assert kL.tobytes() == F.keys.tobytes() and kR.tobytes() == F.keys_right.tobytes()
        sfs = ex.GetScaleFactors()
        Rcw = _rot(0.015, -0.02, 0.01); tcw = np.array([0.2, -0.05, 0.15], np.float32)
        pos, normal, mind, maxd, bad, obs, desc = _scene(F, rng, Rcw, tcw, npts)
        bounds = (0.0, float(w), 0.0, float(h))
        left = views.frame_view(kL, dL, sfs, w, h); right = views.frame_view(kR, dR, sfs, w, h)
        f2 = views.fisheye_frame_view(left, right, F.l2r, F.r2l)
        for th, far in ((1.0, False), (3.0, True)):
            rl, rr, ref_as, ref_n, pose = F.search_local_points(Rcw, tcw, pos, normal, mind, maxd, bad, obs, desc, 0.5, True, th, far, 9.0, 0.8)
            tl, tr, asg, n = M.SearchLocalPointsRig(ex, f2, pose, CAM1, CAM2, bounds, sfs, pos, normal, mind, maxd, bad, obs, desc, 0.5, th, far, 9.0, 0.8)
            il, ir = rl["in_view"], rr["in_view_r"]
            assert 0.15 * npts < il.sum() < 0.97 * npts and 0.15 * npts < ir.sum() < 0.97 * npts and (il != ir).sum() >= 5
            assert np.array_equal(tl["in_view"].astype(bool), il), "mbTrackInView"
            assert np.array_equal(tr["in_view_r"].astype(bool), ir), "mbTrackInViewR"
            for k in ("proj_x", "proj_y", "depth", "view_cos"):
                assert tl[k][il].tobytes() == rl[k][il].tobytes(), k
            for k in ("proj_xr", "proj_yr", "depth_r", "view_cos_r"):
                assert tr[k][ir].tobytes() == rr[k][ir].tobytes(), k
            assert np.array_equal(tl["scale_level"][il], rl["scale_level"][il]) and np.array_equal(tr["scale_level_r"][ir], rr["scale_level_r"][ir])
            assert (tl["scale_level"][~il] == -1).all() and (rl["scale_level"][~il] == -1).all()
            assert n == ref_n and ref_n > npts // 12 and np.array_equal(asg, ref_as), "rig SearchByProjection differs (%d vs %d matches)" % (n, ref_n)
            # the producer alone
            tl2, tr2, _, _ = M.SearchLocalPointsRig(ex, None, pose, CAM1, CAM2, bounds, sfs, pos, normal, mind, maxd, search=False)
            assert np.array_equal(tl2["in_view"], tl["in_view"]) and np.array_equal(tr2["in_view_r"], tr["in_view_r"])
        # one Kannala-Brandt camera alone (k_frustum, camera_type = 1, Nleft == -1 semantics): the left camera of the rig as a monocular frame
        t1, _, _ = M.SearchLocalPoints(ex, None, pose["Rcw"], pose["tcw"], CAM1, bounds, 0.0, sfs, pos, normal, mind, maxd, search=False)
        assert np.array_equal(t1["in_view"].astype(bool), il) and t1["proj_x"][il].tobytes() == rl["proj_x"][il].tobytes() and t1["depth"][il].tobytes() == rl["depth"][il].tobytes()
        ex.close()


def test_local_points_rig_emulated(emu_lib):
    _check(emu_lib, (31,), 1500, 376, 376, 500, (0, 375))


@pytest.mark.gpu
def test_local_points_rig_gpu(hip_lib):
    _check(hip_lib, (31, 32, 33), 5000)
