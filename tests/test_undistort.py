"""Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:1003-1075) on the device: cv::undistortPoints is an OpenCV primitive (restated,
OpenCV is absent: parity unpinned like the other cv:: primitives).  Pins here: the restatement against an independent definition (exact
rational-free check: the distortion model applied to the result gives the input back, and a float64 numpy transcription of the published
iteration agrees bit for bit); the product against the restatement; and the product against the reference's OWN Frame.cc (RGB-D constructor,
UndistortKeyPoints, ComputeStereoFromRGBD, ComputeImageBounds, the grid) over that primitive, with TUM1.yaml's coefficients - mvKeysUn, mvuRight,
the image bounds and the batched SearchLocalPoints on the undistorted keypoints."""
import numpy as np
import pytest

import oracle_lib as ol
from orb_slam3_detailed_comments_amd import ORBextractor, synth
from orb_slam3_detailed_comments_amd import matcher as M

# Examples/RGB-D/TUM1.yaml: Camera.fx .. k3
K_TUM1 = (517.306408, 516.469215, 318.643040, 255.313989)
D_TUM1 = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)
STRICT_SCENES = True          # (the random-coefficient sweep of round 5 switches the "the scene has to be a test" assertions off)


def _numpy_undistort(p, K, d, variant):
    fx, fy, cx, cy = [np.float64(np.float32(v)) for v in K]
    k = [np.float64(np.float32(v)) for v in d] + [np.float64(0)] * (5 - len(d))
    ifx, ify = 1.0 / fx, 1.0 / fy
    out = np.zeros_like(p)
    for i, (u, v) in enumerate(p.astype(np.float64)):
        x = (u - cx) * ifx; y = (v - cy) * ify; x0, y0 = x, y
        for _ in range(5):
            r2 = x * x + y * y
            icdist = 1.0 / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
            if variant == 0 and icdist < 0:
                x = (u - cx) * ifx; y = (v - cy) * ify
                break
            dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x); dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
            x = (x0 - dx) * icdist; y = (y0 - dy) * icdist
        out[i] = (np.float32(fx * x + 0.0 * y + cx), np.float32(0.0 * x + fy * y + cy))
    return out


def _distort(pu, K, d):
    """the forward model (what the lens does to an ideal pixel), float64"""
    fx, fy, cx, cy = K; k1, k2, p1, p2, k3 = d
    x = (pu[:, 0].astype(np.float64) - cx) / fx; y = (pu[:, 1].astype(np.float64) - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x); yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([xd * fx + cx, yd * fy + cy], 1)


def test_undistort_primitive_against_independent_definitions():
    rng = np.random.default_rng(1)
    pts = np.concatenate([rng.uniform(0, 640, (4000, 1)), rng.uniform(0, 480, (4000, 1))], 1).astype(np.float32)
    pts = np.concatenate([pts, np.array([[0, 0], [640, 0], [0, 480], [640, 480], [318.643040, 255.313989]], np.float32)])
    for K, d in ((K_TUM1, D_TUM1), (K_TUM1, D_TUM1[:4]), ((458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05))):
        for variant in (0, 1):
            got = ol.oracle_undistort(pts, K, d, variant)
            assert got.tobytes() == _numpy_undistort(pts, K, d, variant).tobytes()
        d5 = tuple(d) + (0.0,) * (5 - len(d))
        back = _distort(got, K, d5)
        inner = (np.abs(pts[:, 0] - K[2]) < 250) & (np.abs(pts[:, 1] - K[3]) < 200)            # five iterations converge well inside the image
        assert np.abs(back - pts)[inner].max() < 0.25 and np.abs(back - pts)[inner].mean() < 0.01, (np.abs(back - pts)[inner].max(), np.abs(back - pts)[inner].mean())   # five iterations: sub-pixel, not exact
    # extreme distortion: the iteration runs into icdist < 0; the two OpenCV generations differ exactly there
    wild = (-5.0, 0.0, 0.0, 0.0, 0.0)
    a, b = ol.oracle_undistort(pts, K_TUM1, wild, 0), ol.oracle_undistort(pts, K_TUM1, wild, 1)
    assert a.tobytes() == _numpy_undistort(pts, K_TUM1, wild, 0).tobytes() and b.tobytes() == _numpy_undistort(pts, K_TUM1, wild, 1).tobytes() and (a != b).any()


def _product_case(lib, w, h, nf, B, npts, gpu):
    rng = np.random.default_rng(31)
    FX, FY, CX, CY = K_TUM1
    scale = w / 640.0
    K = (FX * scale, FY * scale, CX * scale, CY * scale); BF = 40.0
    imgs = [synth.corner_field(w, h, seed=700 + b, nrect=int(2500 * w * h / (640 * 480))) for b in range(B)]
    depth = (2.0 + np.sin(np.arange(w)[None, :] / 50.0) + np.cos(np.arange(h)[:, None] / 40.0)).astype(np.float32)
    depth[rng.uniform(size=(h, w)) < 0.1] = 0
    refs = [ol.ReferenceFrame(im, None, nf, fx=K[0], fy=K[1], cx=K[2], cy=K[3], bf=BF, depth=depth, dist=D_TUM1) for im in imgs]
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    ex.set_undistort(K, D_TUM1)
    res = ex.extract_batch(np.stack(imgs))
    kun = ex.fetch_undistorted()
    M.ComputeStereoFromRGBD(ex, np.broadcast_to(depth, (B, h, w)).copy(), BF)
    u, dep, _ = M.StereoFetch(ex, B)
    bounds = ex.undistorted_bounds(w, h)
    for b in range(B):
        F = refs[b]
        assert res[b][1].tobytes() == F.keys.tobytes()
        assert kun[b, :F.N].tobytes() == F.keys_un.tobytes(), "mvKeysUn, frame %d" % b
        assert not STRICT_SCENES or (F.keys_un["x"] != F.keys["x"]).mean() > 0.9
        assert u[b, :F.N].tobytes() == F.u_right.tobytes() and dep[b, :F.N].tobytes() == F.depth.tobytes()
        assert np.array_equal(np.float32(bounds), F.bounds[[0, 2, 1, 3]]), (bounds, F.bounds)          # ref_frame_constants: minX, minY, maxX, maxY
    assert not STRICT_SCENES or (bounds[0] != 0.0 and bounds[1] != float(w))
    # the batched SearchLocalPoints on the undistorted keypoints, with the undistorted image bounds, against the reference frames
    from test_local_points import _rot
    sfs = ex.GetScaleFactors()
    pos = np.zeros((npts, 3), np.float32); desc = np.zeros((npts, 32), np.uint8); octv = np.zeros(npts)
    for i in range(npts):
        F = refs[i % B]; j = int(rng.integers(0, F.N))
        z = float(F.depth[j]) if F.depth[j] > 0 else 3.0
        pos[i] = ((F.keys_un["x"][j] - K[2]) / K[0] * z, (F.keys_un["y"][j] - K[3]) / K[1] * z, z); desc[i] = F.desc[j]; octv[i] = F.keys["octave"][j]
        desc[i, int(rng.integers(0, 32))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    dn = np.linalg.norm(pos, axis=1); normal = (pos / dn[:, None]).astype(np.float32)
    maxd = (dn * 1.2 ** octv).astype(np.float32); mind = (maxd / 1.2 ** 7).astype(np.float32)
    obs = rng.uniform(size=npts) < 0.9; bad = rng.uniform(size=npts) < 0.03
    poses = [(_rot(*(rng.normal(0, 0.003, 3))), rng.normal(0, 0.01, 3).astype(np.float32)) for _ in range(B)]
    rp = M.ResidentPoints(ex, pos, normal, mind, maxd, desc)
    lp = M.LocalPointsBatch(ex, rp, B, K, bounds, BF, sfs)
    lp.set_poses(poses)
    lp.enqueue(0, is_bad=bad, has_obs=obs, use_u_right=True, th=3.0, want_in_view=True)
    asg, nm, inv = lp.fetch()
    total = 0
    for b in range(B):
        F = refs[b]
        tr, ref_as, ref_n = F.search_local_points(poses[b][0], poses[b][1], pos, normal, mind, maxd, bad, obs, desc, 0.5, True, 3.0, False, 50.0, 0.8)
        assert np.array_equal(inv[b].astype(bool), tr["in_view"]) and nm[b] == ref_n and np.array_equal(asg[b, :F.N], ref_as), "frame %d" % b
        total += ref_n
    assert not STRICT_SCENES or total > 15 * B
    # switched off again: mvKeysUn = mvKeys
    ex.set_undistort(None)
    ex.extract_batch(np.stack(imgs[:1]))
    assert ex.fetch_undistorted()[0, :refs[0].N].tobytes() == refs[0].keys.tobytes() and ex.undistorted_bounds(w, h) == (0.0, float(w), 0.0, float(h))
    rp.close(); ex.close()


@pytest.mark.skipif(ol.reference_frame_lib() is None, reason="oracle/_ref/libref_frame.so not built (needs /root/reference)")
def test_undistorted_rgbd_frames_emulated(emu_lib):
    _product_case(emu_lib, 320, 240, 300, 2, 600, False)


@pytest.mark.gpu
@pytest.mark.skipif(ol.reference_frame_lib() is None, reason="oracle/_ref/libref_frame.so not built")
def test_undistorted_rgbd_frames_gpu(hip_lib):
    _product_case(hip_lib, 640, 480, 1000, 4, 5000, True)


def test_model_change_after_extraction_is_refused(emu_lib):
    """orbx_set_undistort belongs to the NEXT extraction: mvKeysUn of the previous one was made with the old model (or was never made), so the
    consumers refuse it instead of reading stale / unallocated keypoints.  Both directions: model switched on, model switched off."""
    from orb_slam3_detailed_comments_amd import OrbxError
    K, D = (517.3, 516.5, 318.6, 255.3), (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)
    img = synth.corner_field(320, 240, seed=4, nrect=400)
    depth = np.full((1, 240, 320), 2.0, np.float32)
    ex = ORBextractor(300, 1.2, 8, 20, 7, lib=emu_lib)
    ex.extract_batch(img[None])
    plain = ex.fetch_undistorted()
    ex.set_undistort(K, D)                                  # after the extraction: nothing of that batch may be handed out any more
    with pytest.raises(OrbxError):
        ex.fetch_undistorted()
    with pytest.raises(OrbxError):
        M.ComputeStereoFromRGBD(ex, depth, 40.0)
    ex.extract_batch(img[None])
    un = ex.fetch_undistorted()
    n = ex.counts()[0] if hasattr(ex, "counts") else None
    assert un.tobytes() != plain.tobytes()
    M.ComputeStereoFromRGBD(ex, depth, 40.0)
    ex.set_undistort(None, None)                            # switched off again: the batch still carries the distorted model's mvKeysUn
    with pytest.raises(OrbxError):
        ex.fetch_undistorted()
    ex.extract_batch(img[None])
    assert ex.fetch_undistorted().tobytes() == plain.tobytes()
    ex.close()
