"""Synthetic inputs for the guided-search parity tests (SURVEY.md §8d config 4 and the keyframe searches)."""
import numpy as np

import oracle_lib as ol
from orb_slam3_detailed_comments_amd import synth, views


def frame_from_image(img, nf, rng, occupied_frac=0.1, with_uright=True, mbf=40.0):
    o = ol.OracleExtractor(nf)
    mono, k, d = o.extract(img)
    scales = o.tables()[2][0]
    N = len(k)
    u = None
    if with_uright:     # ComputeStereoFromRGBD-like: uRight = x - mbf/depth for 70 % of the keypoints, -1 otherwise
        depth = rng.uniform(0.5, 8.0, N).astype(np.float32)
        u = np.where(rng.random(N) < 0.7, k["x"] - np.float32(mbf) / depth, np.float32(-1)).astype(np.float32)
    occ = (rng.random(N) < occupied_frac).astype(np.uint8)
    h, w = img.shape
    return views.frame_view(k, d, scales, w, h, u, occ, mbf), k, d, u, scales


def flip_bits(desc, rng, maxflips):
    out = desc.copy()
    for r in range(len(out)):
        nb = int(rng.integers(0, maxflips + 1))
        for b in rng.integers(0, 256, nb):
            out[r, b >> 3] ^= np.uint8(1 << (b & 7))
    return out


def map_points_for_frame(k, d, u, scales, M, rng, w, h):
    """M map points: 60 % are noisy copies of a keypoint (position jitter, 0..40 flipped bits), the rest random."""
    N = len(k)
    src = rng.integers(0, N, M)
    good = rng.random(M) < 0.6
    px = np.where(good, k["x"][src] + rng.uniform(-3, 3, M), rng.uniform(0, w, M)).astype(np.float32)
    py = np.where(good, k["y"][src] + rng.uniform(-3, 3, M), rng.uniform(0, h, M)).astype(np.float32)
    lvl = np.clip(k["octave"][src] + rng.integers(-1, 2, M), 0, len(scales) - 1).astype(np.int32)
    ur = (np.where(u[src] > 0, u[src], px - 5.0) + rng.uniform(-2, 2, M)).astype(np.float32) if u is not None else (px - 5).astype(np.float32)
    desc = np.where(good[:, None], flip_bits(d[src], rng, 40), rng.integers(0, 256, (M, 32), dtype=np.uint8)).astype(np.uint8)
    vc = np.where(rng.random(M) < 0.5, 0.9985, 0.99).astype(np.float32)
    return views.map_point_view((rng.random(M) < 0.9), px, py, ur, lvl, vc, rng.uniform(0.5, 60, M), (rng.random(M) < 0.02),
                                (rng.random(M) < 0.95), desc)


def last_frame_for(k, d, scales, rng, w, h, mbf, shift=(4.0, -2.0)):
    N = len(k)
    valid = rng.random(N) < 0.8
    u = (k["x"] + shift[0] + rng.uniform(-1.5, 1.5, N)).astype(np.float32)
    v = (k["y"] + shift[1] + rng.uniform(-1.5, 1.5, N)).astype(np.float32)
    invz = (1.0 / rng.uniform(0.5, 8.0, N)).astype(np.float32)
    ang = ((k["angle"] + np.where(rng.random(N) < 0.8, rng.uniform(-8, 8, N), rng.uniform(0, 360, N))) % 360).astype(np.float32)
    return views.last_frame_view(valid, u, v, invz, k["octave"], ang, (rng.random(N) < 0.7), flip_bits(d, rng, 30))


def feature_vector(desc, nbits=6):
    """Stand-in for DBoW2's FeatureVector: node id = the first `nbits` bits of the descriptor, CSR with ascending node ids
    and features in insertion (index) order, like std::map<NodeId, vector<unsigned>>."""
    node = (desc[:, 0].astype(np.uint32) & ((1 << nbits) - 1))
    ids = np.unique(node)
    start = [0]; feat = []
    for n in ids:
        f = np.nonzero(node == n)[0]
        feat += f.tolist(); start.append(len(feat))
    return ids.astype(np.uint32), np.array(start, np.int32), np.array(feat, np.uint32)


def keyframe_pair(rng, nf=800, w=640, h=480):
    L, R = synth.stereo_pair(w, h, seed=int(rng.integers(0, 1000)), band=h)   # one disparity: a pure x translation
    out = []
    for img in (L, R):
        o = ol.OracleExtractor(nf)
        mono, k, d = o.extract(img)
        q, um, tabs = o.tables()
        nid, st, ft = feature_vector(d)
        u = np.where(rng.random(len(k)) < 0.3, k["x"] - 3.0, -1.0).astype(np.float32)
        mp = (rng.random(len(k)) < 0.25).astype(np.uint8)
        out.append((views.key_frame_view(k, d, tabs[0], tabs[2], nid, st, ft, u, mp), k, d))
    # fundamental matrix of a pure x translation with identical intrinsics: l2 = (0, -1, y1) up to scale -> y2 == y1
    F12 = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32)
    ep = np.array([1.0e6, h / 2.0], np.float32)
    return out, F12, ep


def projected_points(k, d, u_right, scales, rng, w, h, M=None, maxflips=45):
    """Map points "already projected by the caller" for the Sim3 / key-frame / Fuse searches: 70 % land within a few pixels of a
    keypoint and carry a noisy copy of its descriptor.  M=None: one point per keypoint, in keypoint order (SearchBySim3 /
    SearchByProjection(Frame, KeyFrame) have one map point per key-frame feature)."""
    N = len(k)
    src = np.arange(N) if M is None else rng.integers(0, N, M)
    M = len(src)
    good = rng.random(M) < 0.7
    px = np.where(good, k["x"][src] + rng.uniform(-4, 4, M), rng.uniform(-20, w + 20, M)).astype(np.float32)
    py = np.where(good, k["y"][src] + rng.uniform(-4, 4, M), rng.uniform(-20, h + 20, M)).astype(np.float32)
    lvl = np.clip(k["octave"][src] + rng.integers(0, 2, M), 0, len(scales) - 1).astype(np.int32)
    base_ur = u_right[src] if u_right is not None else px - 4.0
    ur = (np.where(base_ur >= 0, base_ur, px - 4.0) + rng.uniform(-1.5, 1.5, M)).astype(np.float32)
    ang = ((k["angle"][src] + np.where(rng.random(M) < 0.8, rng.uniform(-6, 6, M), rng.uniform(0, 360, M))) % 360).astype(np.float32)
    desc = np.where(good[:, None], flip_bits(d[src], rng, maxflips), rng.integers(0, 256, (M, 32), dtype=np.uint8)).astype(np.uint8)
    return views.projected_point_view(rng.random(M) < 0.85, px, py, lvl, desc, ur, ang)


def shifted_keyframe(k, d, scales, rng, w, h, shift=(6.0, -3.0)):
    """A second key frame seeing the same scene: a shuffled 85 % subset of the keypoints, moved by `shift` plus noise, descriptors
    with up to 30 flipped bits.  Returns (frame_view, keys2, desc2, perm) with perm[j] = index in k of feature j."""
    N = len(k)
    perm = rng.permutation(N)[: int(0.85 * N)]
    k2 = k[perm].copy()
    k2["x"] = (k2["x"] + shift[0] + rng.uniform(-1, 1, len(perm))).astype(np.float32)
    k2["y"] = (k2["y"] + shift[1] + rng.uniform(-1, 1, len(perm))).astype(np.float32)
    d2 = flip_bits(d[perm], rng, 30)
    return views.frame_view(k2, d2, scales, w, h), k2, d2, perm


def fisheye_rig(k, d, scales, rng, w, h, occupied_frac=0.1):
    """A two-camera frame (Nleft != -1): camera 1 = (k, d); camera 2 sees the same scene shifted (shifted_keyframe).  About half of
    the correspondences are known stereo matches (mvLeftToRightMatch / mvRightToLeftMatch), the rest -1."""
    occL = (rng.random(len(k)) < occupied_frac).astype(np.uint8)
    left = views.frame_view(k, d, scales, w, h, None, occL)
    _, k2, d2, perm = shifted_keyframe(k, d, scales, rng, w, h, shift=(-9.0, 1.5))
    occR = (rng.random(len(k2)) < occupied_frac).astype(np.uint8)
    right = views.frame_view(k2, d2, scales, w, h, None, occR)
    l2r = np.full(len(k), -1, np.int32); r2l = np.full(len(k2), -1, np.int32)
    known = rng.random(len(k2)) < 0.5
    for j in np.nonzero(known)[0]:
        l2r[perm[j]] = j; r2l[j] = perm[j]
    return views.fisheye_frame_view(left, right, l2r, r2l), k2, d2, perm


def map_points_right_for(k2, d2, mps, scales, rng, w, h):
    """The *R tracking fields for the map points of `mps`: projections near random right keypoints, some levels unset (-1)."""
    M = mps.view.M; N2 = len(k2)
    src = rng.integers(0, N2, M)
    good = rng.random(M) < 0.6
    pxr = np.where(good, k2["x"][src] + rng.uniform(-3, 3, M), rng.uniform(0, w, M)).astype(np.float32)
    pyr = np.where(good, k2["y"][src] + rng.uniform(-3, 3, M), rng.uniform(0, h, M)).astype(np.float32)
    lvl = np.clip(k2["octave"][src] + rng.integers(-1, 2, M), 0, len(scales) - 1).astype(np.int32)
    lvl[rng.random(M) < 0.05] = -1
    vc = np.where(rng.random(M) < 0.5, 0.9985, 0.99).astype(np.float32)
    # descriptors of the points whose right projection is "good" should resemble the right keypoint: overwrite those rows of mps' descriptors
    desc = mps.keep[9]
    upd = good & (rng.random(M) < 0.5)
    desc[upd] = flip_bits(d2[src[upd]], rng, 40)
    return views.map_point_right_view(rng.random(M) < 0.7, pxr, pyr, lvl, vc)
