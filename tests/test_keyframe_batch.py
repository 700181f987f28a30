"""ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:2196-2324, Tracking::Relocalization) for a batch of
frames on the device (orbm_search_by_projection_keyframe_batch: k_keyframe_queries, k_area_search_threads, k_lastframe_accept) against the REFERENCE:
its own Frame (stereo constructor, SetPose), a key frame with map points, and its own ORBmatcher.cc, called once per frame
(oracle/ref_frame_driver.cpp: ref_frame_search_keyframe), and against the single-frame product call (orbm_search_by_projection_keyframe behind
orbm_project_points).  Key-frame points: missing, bad and already-found ones, points out of their distance range (some exactly on the
0.8 mfMinDistance / 1.2 mfMaxDistance limits' side), duplicated points that compete for a keypoint (any accepted point occupies, observations or
not), pre-occupied keypoints, rotated key-frame keypoints (the histogram takes matches back), th = 10 / ORBdist = 100 and th = 3 / ORBdist = 64
(src/Tracking.cc:4480, :4500)."""
import numpy as np
import pytest

import oracle_lib as ol
from orb_slam3_detailed_comments_amd import ORBextractor, synth, views
from orb_slam3_detailed_comments_amd import matcher as M
from test_local_points import _rot, FX, FY, CX, CY, BF

pytestmark = pytest.mark.skipif(ol.reference_frame_lib() is None, reason="oracle/_ref/libref_frame.so not built (needs /root/reference)")
BASE = 0.110074


# (th, ORBdist, with occupied keypoints, mbCheckOrientation) of the runs; tools/soak_batched_fuzz.py replaces them with random draws
PARAM_SETS = ((10.0, 100, False, True), (3.0, 64, True, True), (10.0, 100, True, False))
STRICT_SCENES = True


def _run(lib, w, h, nf, B, seed=0):
    rng = np.random.default_rng(4242 + B + 1000 * seed)
    pairs = [synth.stereo_pair(w, h, seed=520 + b + 37 * seed, nrect=int(3000 * w * h / (752 * 480))) for b in range(B)]
    refs = [ol.ReferenceFrame(l, r, nf, fx=FX, fy=FY, cx=CX, cy=CY, bf=BF) for l, r in pairs]
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    cap = ex.max_keypoints()
    res = ex.extract_batch(np.stack([l for l, _ in pairs] + [r for _, r in pairs]))
    sfs = ex.GetScaleFactors()
    cam, bounds = (FX, FY, CX, CY), (0.0, float(w), 0.0, float(h))
    capK = cap + 9
    n = np.zeros(B, np.int32); pos = np.zeros((B, capK, 3), np.float32); kind = np.zeros((B, capK), np.uint8)
    mind = np.zeros((B, capK), np.float32); maxd = np.zeros((B, capK), np.float32); angle = np.zeros((B, capK), np.float32); desc = np.zeros((B, capK, 32), np.uint8)
    poses = []
    for b in range(B):
        k, d = res[b][1], res[b][2]; N = len(k)
        assert k.tobytes() == refs[b].keys.tobytes()
        NK = N + 7; n[b] = NK
        R, t = _rot(*(rng.normal(0, 0.01, 3))), rng.normal(0, 0.05, 3).astype(np.float32)
        poses.append((R, t))
        src = rng.integers(0, N, NK)
        z = rng.uniform(1.0, 10.0, NK)
        Xc = np.stack([(k["x"][src] + rng.normal(0, 1.5, NK) - CX) / FX * z, (k["y"][src] + rng.normal(0, 1.5, NK) - CY) / FY * z, z], 1)
        Xc[rng.uniform(size=NK) < 0.03, 2] *= -1.0                                       # behind the camera: this search has no depth test
        Xw = (R.astype(np.float64).T @ (Xc - t.astype(np.float64)).T).T
        pos[b, :NK] = Xw.astype(np.float32)
        Ow = -(R.astype(np.float64).T @ t.astype(np.float64))
        dist = np.linalg.norm(Xw - Ow, axis=1)
        octv = k["octave"][src].astype(np.int64)
        mx = dist * 1.2 ** octv * rng.uniform(0.9, 1.1, NK)                               # PredictScale lands around the source keypoint's level
        mn = mx / 1.2 ** 7
        out = rng.uniform(size=NK) < 0.08                                                # out of range on either side, some just beyond a limit
        mx[out] = dist[out] / 1.2 * rng.choice([0.5, 0.99999, 1.00001], out.sum())
        far = rng.uniform(size=NK) < 0.05
        mn[far] = dist[far] / 0.8 * rng.choice([2.0, 1.00001, 0.99999], far.sum())
        mind[b, :NK] = mn.astype(np.float32); maxd[b, :NK] = mx.astype(np.float32)
        kind[b, :NK] = rng.choice([0, 1, 2, 3], NK, p=[0.1, 0.75, 0.05, 0.1])
        ang = k["angle"][src] + rng.normal(0, 4.0, NK); ang[rng.uniform(size=NK) < 0.15] += rng.uniform(40, 300)
        angle[b, :NK] = np.mod(ang, 360.0).astype(np.float32)
        dd = d[src].copy()
        for i in range(NK):
            for bit in rng.choice(256, int(rng.integers(0, 60)), replace=False):
                dd[i, bit >> 3] ^= np.uint8(1 << (bit & 7))
        desc[b, :NK] = dd
        for i in rng.choice(NK - 70, NK // 5, replace=False):                            # duplicates compete for one keypoint
            j = i + int(rng.choice([1, 2, 63, 64, 65]))
            pos[b, j] = pos[b, i]; desc[b, j] = desc[b, i]; mind[b, j] = mind[b, i]; maxd[b, j] = maxd[b, i]; kind[b, j] = kind[b, i]
    valid = (kind == 1).astype(np.uint8)
    occupied = np.zeros((B, cap), np.uint8)
    for b in range(B):
        occupied[b, rng.choice(refs[b].N, refs[b].N // 8, replace=False)] = 1
    kb = M.KeyFrameBatch(ex, B, cam, bounds, BF, sfs)
    kb.set_poses(poses)
    matcher = M.ORBmatcher(0.9, True)
    for th, orb_dist, use_occ, ori in PARAM_SETS:
        occ = occupied if use_occ else None
        matcher.mbCheckOrientation = ori
        kb.enqueue(n, pos, valid, mind, maxd, angle, desc, th, orb_dist, ori, occ)
        asg, nm = kb.fetch()
        total, resets = 0, 0
        for b in range(B):
            NK = int(n[b]); N = refs[b].N
            ref_n, ref_as = refs[b].search_keyframe(poses[b][0], poses[b][1], pos[b, :NK], kind[b, :NK], mind[b, :NK], maxd[b, :NK], angle[b, :NK], desc[b, :NK], th, orb_dist, ori, 0.9,
                                                    None if occ is None else occ[b, :N])
            assert nm[b] == ref_n and np.array_equal(asg[b, :N], ref_as), "frame %d (th %g) vs the reference: %d vs %d matches" % (b, th, nm[b], ref_n)
            total += ref_n; resets += int((ref_as == -2).sum())
        assert not STRICT_SCENES or (total > 60 * B and (resets > 0) == ori)
    ex.close()


def test_keyframe_batch_emulated(emu_lib):
    _run(emu_lib, 376, 240, 500, 3)


@pytest.mark.gpu
def test_keyframe_batch_gpu(hip_lib):
    _run(hip_lib, 752, 480, 1200, 16)


def test_keyframe_batch_edge_cases(emu_lib):
    """Ragged and empty inputs of orbm_search_by_projection_keyframe_batch: a frame without key-frame points, a frame whose points are all invalid, a frame
    whose points all project outside the image, one frame alone, rows longer than any frame uses; bad arguments are refused before anything runs."""
    from orb_slam3_detailed_comments_amd import OrbxError
    w, h, nf, B = 376, 240, 400, 3
    rng = np.random.default_rng(11)
    pairs = [synth.stereo_pair(w, h, seed=880 + b, nrect=800) for b in range(B)]
    refs = [ol.ReferenceFrame(l, r, nf, fx=FX, fy=FY, cx=CX, cy=CY, bf=BF) for l, r in pairs]
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=emu_lib)
    res = ex.extract_batch(np.stack([l for l, _ in pairs] + [r for _, r in pairs]))
    sfs = ex.GetScaleFactors(); cap = ex.max_keypoints()
    cam, bounds = (FX, FY, CX, CY), (0.0, float(w), 0.0, float(h))
    capK = 700
    n = np.array([0, 300, 500], np.int32)
    pos = np.zeros((B, capK, 3), np.float32); kind = np.zeros((B, capK), np.uint8); mind = np.full((B, capK), 0.1, np.float32); maxd = np.full((B, capK), 50.0, np.float32)
    angle = np.zeros((B, capK), np.float32); desc = rng.integers(0, 256, (B, capK, 32), dtype=np.uint8)
    I, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    poses = [(I, z3)] * B
    k1, d1 = res[1][1], res[1][2]
    src = rng.integers(0, len(k1), 300); zz = rng.uniform(2, 6, 300)
    pos[1, :300] = np.stack([(k1["x"][src] - CX) / FX * zz, (k1["y"][src] - CY) / FY * zz, zz], 1); desc[1, :300] = d1[src]; kind[1, :300] = 2   # all bad: nothing may match
    pos[2, :500] = np.stack([rng.uniform(50, 90, 500), rng.uniform(50, 90, 500), rng.uniform(1, 2, 500)], 1); kind[2, :500] = 1                     # far outside the image
    valid = (kind == 1).astype(np.uint8)
    kb = M.KeyFrameBatch(ex, B, cam, bounds, BF, sfs); kb.set_poses(poses)
    kb.enqueue(n, pos, valid, mind, maxd, angle, desc, 10.0, 100, True, None)
    asg, nm = kb.fetch()
    assert nm.tolist() == [0, 0, 0] and (asg[:, :cap] == -1).all()
    for b in range(B):
        ref_n, ref_as = refs[b].search_keyframe(I, z3, pos[b, :n[b]].reshape(-1, 3), kind[b, :n[b]], mind[b, :n[b]], maxd[b, :n[b]], angle[b, :n[b]], desc[b, :n[b]].reshape(-1, 32), 10.0, 100)
        assert ref_n == 0 and (ref_as == -1).all()
    # the same points, good this time, on frame 1 alone (first = 1, B = 1): matches, and equal to the reference
    kind1 = np.ones((1, capK), np.uint8)
    one = M.KeyFrameBatch(ex, 1, cam, bounds, BF, sfs); one.set_poses([(I, z3)])
    one.enqueue(n[1:2], pos[1:2], kind1, mind[1:2], maxd[1:2], angle[1:2], desc[1:2], 10.0, 100, False, None, first=1)
    a1, m1 = one.fetch()
    ref_n, ref_as = refs[1].search_keyframe(I, z3, pos[1, :300], kind1[0, :300], mind[1, :300], maxd[1, :300], angle[1, :300], desc[1, :300], 10.0, 100, False)
    assert m1[0] == ref_n > 20 and np.array_equal(a1[0, :refs[1].N], ref_as)
    # refusals
    with pytest.raises(OrbxError):
        one.enqueue(n[1:2], pos[1:2], kind1, mind[1:2], maxd[1:2], angle[1:2], desc[1:2], 10.0, 100, False, None, first=2 * B)        # frames beyond the extraction
    with pytest.raises(OrbxError):
        one.enqueue(np.array([capK + 1], np.int32), pos[1:2], kind1, mind[1:2], maxd[1:2], angle[1:2], desc[1:2], 10.0, 100, False, None, first=1)   # more points than rows
    with pytest.raises(OrbxError):
        one.enqueue(n[1:2], pos[1:2], kind1, mind[1:2], maxd[1:2], angle[1:2], desc[1:2], 10.0, 300, False, None, first=1)             # ORBdist > 256
    with pytest.raises(OrbxError):
        one.fetch()                                                                                                                  # a refused call leaves nothing to fetch
    ex.close()
