"""Batched, device-resident Tracking::SearchLocalPoints (orbm_search_local_points_batch: k_grid_build / k_frustum / k_area_search over B frames,
k_local_accept = the sequential accept loop of ORBmatcher::SearchByProjection on the device) and Frame::ComputeStereoFromRGBD
(orbm_stereo_from_depth), against the reference's own Frame.cc + ORBmatcher.cc run once per frame (oracle/_ref/libref_frame.so): stereo
frames (uRight from the device stereo matcher) and RGB-D frames (uRight from the depth image), several poses, occupied keypoints, map points
without observations, bad points.  Bar: identical assignments, match counts and mbTrackInView for every frame; uRight / depth bit-identical."""
import numpy as np
import pytest

import oracle_lib as ol
from orb_slam3_detailed_comments_amd import ORBextractor, synth
from orb_slam3_detailed_comments_amd import matcher as M
from test_local_points import _rot, _scene, FX, FY, CX, CY, BF

pytestmark = pytest.mark.skipif(ol.reference_frame_lib() is None, reason="oracle/_ref/libref_frame.so not built (needs /root/reference)")
BASE = 0.110074


def _depth_image(w, h, seed):
    rng = np.random.default_rng(900 + seed)
    yy, xx = np.mgrid[0:h, 0:w]
    d = (2.5 + 1.5 * np.sin(xx / 37.0 + seed) * np.cos(yy / 23.0) + rng.normal(0, 0.02, (h, w))).astype(np.float32)
    holes = rng.uniform(size=(h, w)) < 0.12          # missing depth: 0, and a few negative readings
    d[holes] = 0.0
    d[rng.uniform(size=(h, w)) < 0.01] = -1.0
    return d


# (th, bFarPoints, with occupied keypoints, viewing-cosine limit, thFarPoints, nnratio) of the runs; tools/soak_batched_fuzz.py replaces them with random draws
PARAM_SETS = ((1.0, False, False, 0.5, 9.0, 0.8), (3.0, True, True, 0.5, 9.0, 0.8))
STRICT_SCENES = True


def _run(lib, w, h, nf, P, npts, rgbd):
    rng = np.random.default_rng(4242 + P + (7 if rgbd else 0))
    pairs = [synth.stereo_pair(w, h, seed=60 + (0 if p < 2 else p), nrect=int(3000 * w * h / (752 * 480))) for p in range(P)]     # frames 0 and 1 show the same scene
    depths = [_depth_image(w, h, p) for p in range(P)]
    refs = [ol.ReferenceFrame(l, r, nf, fx=FX, fy=FY, cx=CX, cy=CY, bf=BF, depth=depths[p] if rgbd else None) for p, (l, r) in enumerate(pairs)]
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    cap = ex.max_keypoints()
    if rgbd:
        res = ex.extract_batch(np.stack([l for l, _ in pairs]))
        M.ComputeStereoFromRGBD(ex, np.stack(depths), BF)
        u, dep, nvalid = M.StereoFetch(ex, P)
        for p in range(P):
            F = refs[p]
            assert res[p][1].tobytes() == F.keys.tobytes()
            assert u[p, :F.N].tobytes() == F.u_right.tobytes() and dep[p, :F.N].tobytes() == F.depth.tobytes(), "ComputeStereoFromRGBD, frame %d" % p
            assert nvalid[p] == int((F.depth > 0).sum()) and 0.5 * F.N < nvalid[p] < F.N
    else:
        res = ex.extract_batch(np.stack([l for l, _ in pairs] + [r for _, r in pairs]))
        lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, P, P, BF, BASE))
        u, dep, _ = M.StereoFetch(ex, P)
        for p in range(P):
            assert u[p, :refs[p].N].tobytes() == refs[p].u_right.tobytes()
    sfs = ex.GetScaleFactors()
    poses = [(_rot(0.02, -0.03, 0.01), np.array([0.3, -0.1, 0.25], np.float32)), (_rot(0.021, -0.028, 0.012), np.array([0.28, -0.11, 0.27], np.float32)),
             (_rot(-0.01, 0.02, 0.0), np.array([-0.2, 0.05, 0.1], np.float32))][:P]
    pos, normal, mind, maxd, bad, obs, desc = _scene(refs[0], rng, poses[0][0], poses[0][1], npts)
    # duplicated map points compete for one keypoint - next to each other (inside one group of 64 of the accept kernel: the optimistic decisions
    # collide and are redone) and far apart; some of the winners have no observations (they do not occupy, a later point overwrites them)
    for i in rng.choice(npts - 70, npts // 4, replace=False):
        j = i + int(rng.choice([1, 2, 5, 63, 64, 65]))
        pos[j] = pos[i]; normal[j] = normal[i]; mind[j] = mind[i]; maxd[j] = maxd[i]; desc[j] = desc[i]
        if rng.uniform() < 0.5:
            desc[j, int(rng.integers(0, 32))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    rp = M.ResidentPoints(ex, pos, normal, mind, maxd, desc)
    occupied = np.zeros((P, cap), np.uint8)
    for p in range(P):
        occupied[p, rng.choice(refs[p].N, refs[p].N // 6, replace=False)] = 1
    lp = M.LocalPointsBatch(ex, rp, P, (FX, FY, CX, CY), (0.0, float(w), 0.0, float(h)), BF, sfs)
    lp.set_poses(poses)
    for th, far, use_occ, cosl, thfar, ratio in PARAM_SETS:
        occ = occupied if use_occ else None
        lp.enqueue(0, is_bad=bad, has_obs=obs, occupied=occ, use_u_right=True, viewing_cos_limit=cosl, th=th, far_points=far, th_far=thfar, nnratio=ratio, want_in_view=True)
        asg, nm, inv = lp.fetch()
        total = 0
        for p in range(P):
            F = refs[p]
            # the reference has no "occupied" input: pre-occupied keypoints are modelled by the single-frame product call, which was pinned against
            # the reference (tests/test_local_points.py); without occupancy the reference itself is the checker
            if occ is None:
                ref_tr, ref_as, ref_n = F.search_local_points(poses[p][0], poses[p][1], pos, normal, mind, maxd, bad, obs, desc, cosl, True, th, far, thfar, ratio)
                assert np.array_equal(inv[p].astype(bool), ref_tr["in_view"]), "mbTrackInView, frame %d" % p
            else:
                from orb_slam3_detailed_comments_amd import views
                fv = views.frame_view(res[p][1], res[p][2], sfs, w, h, u_right=u[p, :F.N], mbf=BF, occupied=occ[p, :F.N])
                _, ref_as, ref_n = M.SearchLocalPoints(ex, fv, poses[p][0], poses[p][1], (FX, FY, CX, CY), (0.0, float(w), 0.0, float(h)), BF, sfs, pos, normal, mind, maxd, bad, obs, desc,
                                                       cosl, th, far, thfar, ratio)
            assert nm[p] == ref_n and np.array_equal(asg[p, :F.N], ref_as) and (asg[p, F.N:] == -1).all(), "frame %d: %d vs %d matches" % (p, nm[p], ref_n)
            total += ref_n
        assert not STRICT_SCENES or total > npts // 20
    # monocular frames (no uRight): the right-coordinate gate is off
    lp.enqueue(0, is_bad=bad, has_obs=obs, use_u_right=False, th=3.0)
    asg, nm, _ = lp.fetch()
    from orb_slam3_detailed_comments_amd import views
    for p in range(P):
        F = refs[p]
        fv = views.frame_view(res[p][1], res[p][2], sfs, w, h)
        _, ref_as, ref_n = M.SearchLocalPoints(ex, fv, poses[p][0], poses[p][1], (FX, FY, CX, CY), (0.0, float(w), 0.0, float(h)), BF, sfs, pos, normal, mind, maxd, bad, obs, desc, 0.5, 3.0)
        assert nm[p] == ref_n and np.array_equal(asg[p, :F.N], ref_as)
    rp.close(); ex.close()


@pytest.mark.parametrize("rgbd", [False, True])
def test_local_points_batch_emulated(emu_lib, rgbd):
    _run(emu_lib, 376, 240, 500, 3, 900, rgbd)


@pytest.mark.gpu
@pytest.mark.parametrize("rgbd", [False, True])
def test_local_points_batch_gpu(hip_lib, rgbd):
    _run(hip_lib, 640, 480, 1000, 3, 5000, rgbd)
    _run(hip_lib, 752, 480, 1200, 3, 3000, rgbd)


def _large_batch(lib, w, h, nf, B, npts, nscenes):
    """B frames x npts points in one batch (the bench's shape) against the single-frame product call, which is pinned to the reference Frame above:
    per-frame offsets of the grid, the queries, the candidate pool and the accept kernel at a size where every workgroup row is populated"""
    from orb_slam3_detailed_comments_amd import views
    rng = np.random.default_rng(77)
    imgs = np.stack([synth.corner_field(w, h, seed=500 + (b % nscenes), nrect=int(2500 * w * h / (640 * 480))) for b in range(B)])      # nscenes distinct scenes, several poses each
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    res = ex.extract_batch(imgs)
    depth = (2.0 + np.sin(np.arange(w)[None, :] / 50.0) + np.cos(np.arange(h)[:, None] / 40.0)).astype(np.float32)
    depth[rng.uniform(size=(h, w)) < 0.1] = 0
    M.ComputeStereoFromRGBD(ex, np.broadcast_to(depth, (B, h, w)).copy(), 40.0)
    u, dep, _ = M.StereoFetch(ex, B)
    sfs = ex.GetScaleFactors()
    cam, bounds = (FX, FY, CX, CY), (0.0, float(w), 0.0, float(h))
    # map points on the rays of keypoints of the 12 scenes, at the depth the depth image gives them
    pos = np.zeros((npts, 3), np.float32); desc = np.zeros((npts, 32), np.uint8); octv = np.zeros(npts)
    for i in range(npts):
        k, d = res[i % nscenes][1], res[i % nscenes][2]
        j = int(rng.integers(0, len(k)))
        z = float(depth[int(k["y"][j]), int(k["x"][j])]) or 3.0
        pos[i] = ((k["x"][j] - CX) / FX * z, (k["y"][j] - CY) / FY * z, z); desc[i] = d[j]; octv[i] = k["octave"][j]
        desc[i, int(rng.integers(0, 32))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    dn = np.linalg.norm(pos, axis=1); normal = (pos / dn[:, None]).astype(np.float32)
    maxd = (dn * 1.2 ** octv).astype(np.float32); mind = (maxd / 1.2 ** 7).astype(np.float32)
    obs = rng.uniform(size=npts) < 0.9; bad = rng.uniform(size=npts) < 0.03
    poses = [(_rot(*(rng.normal(0, 0.004, 3))), rng.normal(0, 0.01, 3).astype(np.float32)) for _ in range(B)]
    rp = M.ResidentPoints(ex, pos, normal, mind, maxd, desc)
    lp = M.LocalPointsBatch(ex, rp, B, cam, bounds, 40.0, sfs)
    lp.set_poses(poses)
    lp.enqueue(0, is_bad=bad, has_obs=obs, use_u_right=True, th=3.0, want_in_view=True)
    asg, nm, inv = lp.fetch()
    total = 0
    for b in range(B):
        n = len(res[b][1])
        fv = views.frame_view(res[b][1], res[b][2], sfs, w, h, u_right=u[b, :n], mbf=40.0)
        tr, ref_as, ref_n = M.SearchLocalPoints(ex, fv, poses[b][0], poses[b][1], cam, bounds, 40.0, sfs, pos, normal, mind, maxd, bad, obs, desc, 0.5, 3.0, False, 50.0, 0.8)
        assert nm[b] == ref_n and np.array_equal(asg[b, :n], ref_as) and np.array_equal(inv[b], tr["in_view"]), "frame %d" % b
        total += ref_n
    assert total > 20 * B
    rp.close(); ex.close()


def test_local_points_many_frames_emulated(emu_lib):
    _large_batch(emu_lib, 320, 240, 300, 7, 800, 3)


@pytest.mark.gpu
def test_local_points_large_batch_gpu(hip_lib):
    _large_batch(hip_lib, 640, 480, 1000, 48, 5000, 12)


def test_candidate_pool_overflow_is_retried(emu_lib):
    """a search window so wide that the candidate pool of a fresh handle overflows: the fetch reports ORBX_E_CAPACITY once (the pool is enlarged),
    the wrapper runs the batch again, and the results are those of the single-frame call"""
    import ctypes as C
    from orb_slam3_detailed_comments_amd import views
    w, h, nf, npts = 320, 240, 300, 400
    rng = np.random.default_rng(11)
    imgs = np.stack([synth.corner_field(w, h, seed=40 + b, nrect=700) for b in range(2)])
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=emu_lib)
    res = ex.extract_batch(imgs)
    k, d = res[0][1], res[0][2]
    src = rng.integers(0, len(k), npts); z = rng.uniform(1, 6, npts)
    pos = np.stack([(k["x"][src] - CX) / FX * z, (k["y"][src] - CY) / FY * z, z], 1).astype(np.float32)
    dn = np.linalg.norm(pos, axis=1); normal = (pos / dn[:, None]).astype(np.float32)
    maxd = (dn * 1.2 ** k["octave"][src]).astype(np.float32); mind = (maxd / 1.2 ** 7).astype(np.float32)
    desc = d[src].copy()
    sfs = ex.GetScaleFactors(); cam, bounds = (FX, FY, CX, CY), (0.0, float(w), 0.0, float(h))
    rp = M.ResidentPoints(ex, pos, normal, mind, maxd, desc)
    lp = M.LocalPointsBatch(ex, rp, 2, cam, bounds, 0.0, sfs)
    poses = [(np.eye(3, dtype=np.float32), np.zeros(3, np.float32))] * 2
    lp.set_poses(poses)
    lp.enqueue(0, use_u_right=False, th=30.0)
    rc = emu_lib.L.orbm_search_local_points_fetch(ex._h, lp.assigned.ctypes.data, lp.cap, lp.nm.ctypes.data, None)
    assert rc == -4                                         # ORBX_E_CAPACITY: tens of thousands of candidates against a pool sized for six per point
    lp.enqueue(0, use_u_right=False, th=30.0)
    asg, nm, _ = lp.fetch()
    for b in range(2):
        fv = views.frame_view(res[b][1], res[b][2], sfs, w, h)
        _, ref_as, ref_n = M.SearchLocalPoints(ex, fv, poses[b][0], poses[b][1], cam, bounds, 0.0, sfs, pos, normal, mind, maxd, None, None, desc, 0.5, 30.0)
        assert nm[b] == ref_n and np.array_equal(asg[b, :len(res[b][1])], ref_as)
    assert nm[0] > 50
    rp.close(); ex.close()
