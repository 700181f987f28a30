"""Synthetic SLAM worlds for the ORBmatcher drop-in check (tests/test_matcher_reference.py).

Run as a script it builds one seeded world in a world-driver library (tests/cpp/matcher_world_driver.cpp: oracle/_ref/libmw_ref.so = the
reference's own ORBmatcher.cc, oracle/_ref/libmw_facade.so = include/orb_slam3_amd/ORBmatcher.h on top of the C ABI), runs every
ORBmatcher method on it and writes everything the methods returned or modified to an .npz.  A separate process per library keeps the
HIP build and the CPU emulator build of the product library out of each other's symbol space.

    python tests/matcher_world.py <driver.so> <orbx library or ''> <seed> <variant> <out.npz>
"""
import ctypes as C
import sys

import numpy as np

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
W, H = 752, 480
FX, FY, CX, CY = 435.2, 435.2, 367.4, 252.2
NLEVELS, SCALE = 8, 1.2
BF = 47.9


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)


class Driver:
    def __init__(self, path, orbx=None):
        if orbx:
            self._orbx = C.CDLL(orbx, mode=C.RTLD_GLOBAL)
        self.L = L = C.CDLL(path)
        L.mw_create.restype = C.c_void_p
        L.mw_flavour.restype = C.c_char_p
        self.w = C.c_void_p(L.mw_create())

    def close(self):
        self.L.mw_destroy(self.w)

    def camera(self, fx=FX, fy=FY, cx=CX, cy=CY):
        return self.L.mw_add_camera(self.w, C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy))

    def mappoint(self, pos, normal, min_dist, max_dist, desc, bad=0, n_obs=1):
        pos = np.ascontiguousarray(pos, np.float32); normal = np.ascontiguousarray(normal, np.float32); desc = np.ascontiguousarray(desc, np.uint8)
        return self.L.mw_add_mappoint(self.w, _p(pos), _p(normal), C.c_float(min_dist), C.c_float(max_dist), _p(desc), int(bad), int(n_obs))

    def track(self, mp, in_view, in_view_r, level, level_r, t7):
        t = np.ascontiguousarray(t7, np.float32)
        self.L.mw_set_track(self.w, int(mp), int(in_view), int(in_view_r), int(level), int(level_r), _p(t))

    def frame(self, keyframe, keys_un, desc, u_right, R, t, cam, cam2=-1, keys_right=None, trl=None, mbf=BF, mb=0.11):
        """one camera: keys_un [N]; rig: keys_un = left keys, keys_right given, desc = left rows then right rows"""
        n_right = -1 if keys_right is None else len(keys_right)
        N = len(keys_un) + max(n_right, 0)
        desc = np.ascontiguousarray(desc, np.uint8); assert desc.shape == (N, 32)
        R = np.ascontiguousarray(R, np.float32); t = np.ascontiguousarray(t, np.float32)
        bounds = np.array([0, 0, W, H], np.float32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        trl_R = trl_t = None
        if trl is not None:
            trl_R = np.ascontiguousarray(trl[0], np.float32); trl_t = np.ascontiguousarray(trl[1], np.float32)
        keys_un = np.ascontiguousarray(keys_un); kr = None if keys_right is None else np.ascontiguousarray(keys_right)
        return self.L.mw_add_frame(self.w, int(keyframe), N, _p(keys_un), _p(keys_un), n_right, _p(kr), _p(desc), _p(ur), _p(R), _p(t), _p(trl_R), _p(trl_t),
                                   _p(bounds), NLEVELS, C.c_float(SCALE), cam, cam2, C.c_float(mbf), C.c_float(mb))

    def set_map_points(self, keyframe, fid, ids, outlier=None):
        ids = np.ascontiguousarray(ids, np.int32); o = None if outlier is None else np.ascontiguousarray(outlier, np.uint8)
        self.L.mw_set_map_points(self.w, int(keyframe), fid, _p(ids), _p(o))

    def get_map_points(self, keyframe, fid, n):
        out = np.zeros(n, np.int32); self.L.mw_get_map_points(self.w, int(keyframe), fid, _p(out)); return out

    def set_feat_vec(self, keyframe, fid, nodes_of_feature):
        """nodes_of_feature[i] = vocabulary node of feature i (DBoW2 adds features in index order)"""
        nodes = np.unique(nodes_of_feature)
        order = np.argsort(nodes_of_feature, kind="stable")
        counts = np.array([(nodes_of_feature == n).sum() for n in nodes])
        start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        self.L.mw_set_feat_vec(self.w, int(keyframe), fid, len(nodes), _p(nodes.astype(np.uint32)), _p(start), _p(order.astype(np.uint32)))

    def mappoint_state(self, mp, kf):
        out = np.zeros(4, np.int32); self.L.mw_get_mappoint_state(self.w, int(mp), int(kf), _p(out)); return out


class Scene:
    """3-D points, descriptors, vocabulary nodes; observations of them from given poses."""

    def __init__(self, rng, npts=900):
        self.rng = rng
        self.X = np.stack([rng.uniform(-5, 5, npts), rng.uniform(-3, 3, npts), rng.uniform(2.5, 14, npts)], 1).astype(np.float32)
        self.base = rng.integers(0, 256, (npts, 32), dtype=np.uint8)
        self.node = rng.integers(0, 60, npts)
        self.angle0 = rng.uniform(0, 360, npts)
        self.level0 = np.clip(np.round(np.log(14.0 / self.X[:, 2]) / np.log(SCALE)).astype(int) - rng.integers(0, 2, npts), 0, NLEVELS - 1)

    def noisy_desc(self, idx, maxflips):
        d = self.base[idx].copy()
        for r in range(len(idx)):
            k = int(self.rng.integers(0, maxflips + 1))
            bits = self.rng.choice(256, k, replace=False)
            for b in bits:
                d[r, b >> 3] ^= np.uint8(1 << (b & 7))
        return d

    def observe(self, R, t, frac=0.8, clutter=250, px_noise=1.5, maxflips=45, rot_off=0.0, stereo_frac=0.6, angle_outliers=0.08, level0_frac=0.0, proj=None):
        """Returns keys (KP array), desc, u_right, point index per feature (-1 clutter), node per feature.  proj: camera projection of the
        camera-frame points (default: the pinhole FX, FY, CX, CY)."""
        rng = self.rng
        Xc = (R.astype(np.float64) @ self.X.T.astype(np.float64)).T + t.astype(np.float64)
        z = Xc[:, 2]
        u = FX * Xc[:, 0] / z + CX; v = FY * Xc[:, 1] / z + CY
        if proj is not None:
            u, v = proj(Xc)
        vis = (z > 0.5) & (u > 8) & (u < W - 8) & (v > 8) & (v < H - 8) & (rng.uniform(size=len(z)) < frac)
        idx = np.nonzero(vis)[0]
        rng.shuffle(idx)
        n = len(idx) + clutter
        keys = np.zeros(n, KP)
        keys["x"][:len(idx)] = u[idx] + rng.normal(0, px_noise, len(idx)); keys["y"][:len(idx)] = v[idx] + rng.normal(0, px_noise, len(idx))
        keys["x"][len(idx):] = rng.uniform(4, W - 4, clutter); keys["y"][len(idx):] = rng.uniform(4, H - 4, clutter)
        lev = np.clip(np.round(np.log(14.0 / z[idx]) / np.log(SCALE)).astype(int) - rng.integers(0, 2, len(idx)), 0, NLEVELS - 1)
        keys["octave"][:len(idx)] = lev; keys["octave"][len(idx):] = rng.integers(0, NLEVELS, clutter)
        if level0_frac > 0:
            keys["octave"][rng.uniform(size=n) < level0_frac] = 0
        ang = (self.angle0[idx] + rot_off + rng.normal(0, 3.0, len(idx))) % 360.0
        out = rng.uniform(size=len(idx)) < angle_outliers
        ang[out] = rng.uniform(0, 360, out.sum())
        keys["angle"][:len(idx)] = ang; keys["angle"][len(idx):] = rng.uniform(0, 360, clutter)
        keys["size"] = 31.0 * SCALE ** keys["octave"]; keys["response"] = rng.uniform(20, 120, n); keys["class_id"] = -1
        desc = np.concatenate([self.noisy_desc(idx, maxflips), rng.integers(0, 256, (clutter, 32), dtype=np.uint8)])
        ur = np.full(n, -1.0, np.float32)
        st = rng.uniform(size=len(idx)) < stereo_frac
        ur[:len(idx)][st] = (keys["x"][:len(idx)][st] - BF / z[idx][st]).astype(np.float32)
        pt = np.concatenate([idx, np.full(clutter, -1)])
        node = np.concatenate([self.node[idx], rng.integers(0, 60, clutter)])
        # shuffle features so that point order and feature order are unrelated
        perm = rng.permutation(n)
        return keys[perm], desc[perm], ur[perm], pt[perm], node[perm]


FUZZ = None         # variant "fuzz": a generator that replaces the methods' parameters (th, ratios, windows) by random draws - the same draws for both flavours


def P(default, choices):
    return default if FUZZ is None else type(default)(FUZZ.choice(choices))


def build_and_run(drv, seed, variant):
    """Runs every method on a world derived from (seed, variant); returns {name: array}."""
    rng = np.random.default_rng(seed)
    out = {}
    L = drv.L
    sc = Scene(rng)
    cam = drv.camera()
    dense = variant == "dense"
    flips = 70 if variant == "hard" else 45
    # map points: one per scene point
    cams_center = np.zeros(3)
    mp_ids = []
    for i in range(len(sc.X)):
        PO = sc.X[i] - cams_center
        d = float(np.linalg.norm(PO))
        maxd = d * SCALE ** int(sc.level0[i]); mind = maxd / SCALE ** (NLEVELS - 1)
        nrm = PO / d + rng.normal(0, 0.15, 3)
        nrm /= np.linalg.norm(nrm)
        if rng.uniform() < 0.05:
            nrm = -nrm                                   # fails the viewing-angle gate
        mp_ids.append(drv.mappoint(sc.X[i] + rng.normal(0, 0.01, 3), nrm, mind, maxd, sc.noisy_desc(np.array([i]), 20)[0], bad=rng.uniform() < 0.04,
                                   n_obs=int(rng.integers(0, 6))))
    mp_ids = np.array(mp_ids)

    I3 = np.eye(3, dtype=np.float32)
    poses = [(I3, np.zeros(3, np.float32)),
             (rot(0.01, -0.02, 0.03), np.array([0.25, -0.03, 0.35], np.float32)),
             (rot(-0.015, 0.03, 0.26), np.array([-0.3, 0.05, -0.30], np.float32))]
    clutter = 600 if dense else 250

    def make(keyframe, pose, rot_off, assoc_frac, **kw):
        R, t = pose
        keys, desc, ur, pt, node = sc.observe(R, t, clutter=clutter, rot_off=rot_off, maxflips=flips, **kw)
        fid = drv.frame(keyframe, keys, desc, ur, R, t, cam)
        ids = np.where((pt >= 0) & (rng.uniform(size=len(pt)) < assoc_frac), mp_ids[np.maximum(pt, 0)], -1).astype(np.int32)
        drv.set_feat_vec(keyframe, fid, node)
        return fid, keys, desc, ur, pt, ids

    def i32(n, fill=-1):
        return np.full(max(n, 1), fill, np.int32)

    # ---- SearchByProjection(Frame, vector<MapPoint*>) : local map tracking --------------------------------
    f0, k0, d0, u0, pt0, ids0 = make(False, poses[1], 0.0, 0.35)
    drv.set_map_points(False, f0, ids0)
    R, t = poses[1]
    Xc = (R @ sc.X.T).T + t
    for i in range(len(sc.X)):
        z = Xc[i, 2]
        uu = FX * Xc[i, 0] / z + CX + rng.normal(0, 1.0); vv = FY * Xc[i, 1] / z + CY + rng.normal(0, 1.0)
        inv = z > 0.5 and 0 <= uu < W and 0 <= vv < H and rng.uniform() < 0.9
        lvl = int(np.clip(sc.level0[i] + rng.integers(-1, 2), 0, NLEVELS - 1))
        drv.track(mp_ids[i], inv, 0, lvl, -1, [uu, vv, uu - BF / z, rng.choice([0.9985, 0.99, 0.95]), z, 0, 1])
    sel = rng.permutation(len(mp_ids))[:700]
    for far in (0, 1):
        drv.set_map_points(False, f0, ids0)
        n = L.mw_search_by_projection_mappoints(drv.w, f0, _p(mp_ids[sel].astype(np.int32)), len(sel), C.c_float(P(3.0 if far else 1.0, [0.5, 1.0, 2.0, 3.0, 6.0])), far, C.c_float(P(9.0, [3.0, 9.0, 30.0])), C.c_float(P(0.8, [0.6, 0.8, 0.95])), 1)
        out["sbp_mappoints_%d" % far] = np.concatenate([[n], drv.get_map_points(False, f0, len(k0))])

    # ---- SearchByProjection(CurrentFrame, LastFrame) : motion model ----------------------------------------
    for tag, mono, pose_c in (("fwd", 0, 1), ("bwd", 0, 2), ("mono", 1, 1)):
        fl, kl, dl, ul, ptl, idsl = make(False, poses[0], 0.0, 0.8)
        drv.set_map_points(False, fl, idsl, outlier=(rng.uniform(size=len(kl)) < 0.05))
        fc, kc, dc, uc, ptc, idsc = make(False, poses[pose_c], 8.0, 0.1)
        drv.set_map_points(False, fc, idsc)
        for th in (7.0, 15.0):
            drv.set_map_points(False, fc, idsc)
            n = L.mw_search_by_projection_frame(drv.w, fc, fl, C.c_float(P(th, [1.0, 3.0, 7.0, 15.0, 30.0])), mono, C.c_float(P(0.9, [0.7, 0.9, 1.0])), 1)
            out["sbp_frame_%s_%d" % (tag, int(th))] = np.concatenate([[n], drv.get_map_points(False, fc, len(kc))])
        drv.set_map_points(False, fc, idsc)
        n = L.mw_search_by_projection_frame(drv.w, fc, fl, C.c_float(P(15.0, [2.0, 15.0, 40.0])), mono, C.c_float(0.9), 0)
        out["sbp_frame_%s_noori" % tag] = np.concatenate([[n], drv.get_map_points(False, fc, len(kc))])

    # ---- key frames ---------------------------------------------------------------------------------------------
    kA, kkA, dA, uA, ptA, idsA = make(True, poses[0], 0.0, 0.7)
    drv.set_map_points(True, kA, idsA)
    kB, kkB, dB, uB, ptB, idsB = make(True, poses[1], 12.0, 0.7)
    drv.set_map_points(True, kB, idsB)

    # ---- SearchByProjection(CurrentFrame, KeyFrame, sAlreadyFound) : relocalisation -----------------------
    fr, kr, dr_, ur_, ptr_, idsr = make(False, poses[1], 10.0, 0.15)
    found = idsr[idsr >= 0][::2].astype(np.int32)
    for th, od in ((10.0, 100), (3.0, 64)):
        drv.set_map_points(False, fr, idsr)
        n = L.mw_search_by_projection_keyframe(drv.w, fr, kA, _p(found), len(found), C.c_float(P(th, [1.0, 3.0, 6.0, 10.0, 25.0])), P(od, [30, 50, 64, 100, 160]), C.c_float(P(0.9, [0.7, 0.9])), 1)
        out["sbp_keyframe_%d" % od] = np.concatenate([[n], drv.get_map_points(False, fr, len(kr))])

    # ---- SearchByProjection(KeyFrame, Sim3, ...) : loop detection / place recognition ----------------------
    Rs, ts = poses[1]
    for s in (1.0, 1.07):
        for with_kfs in (0, 1):
            pts = rng.permutation(len(mp_ids))[:600].astype(np.int32)
            matched = np.where(rng.uniform(size=len(kkB)) < 0.1, idsB, -1).astype(np.int32)
            mkf = i32(len(kkB))
            pkf = rng.integers(0, 2, len(pts)).astype(np.int32)
            n = L.mw_search_by_projection_sim3(drv.w, kB, C.c_float(s), _p(Rs), _p((ts * s).astype(np.float32)), _p(mp_ids[pts].astype(np.int32)), len(pts), _p(pkf), with_kfs,
                                               _p(matched), _p(mkf), P(8, [1, 3, 8, 14]), C.c_float(P(1.0 if with_kfs else 1.5, [0.7, 1.0, 1.5])))
            out["sbp_sim3_%d_%d" % (int(s * 100), with_kfs)] = np.concatenate([[n], matched, mkf])

    # ---- SearchByBoW -----------------------------------------------------------------------------------------------
    fb, kb, db, ub, ptb, idsb = make(False, poses[1], 20.0, 0.0)
    for ori in (1, 0):
        o = i32(len(kb))
        n = L.mw_search_by_bow_frame(drv.w, kA, fb, _p(o), C.c_float(P(0.75, [0.55, 0.75, 0.9, 1.0])), ori)
        out["bow_frame_%d" % ori] = np.concatenate([[n], o])
        o = i32(len(kkA))
        n = L.mw_search_by_bow_keyframes(drv.w, kA, kB, _p(o), C.c_float(P(0.8, [0.55, 0.8, 0.95])), ori)
        out["bow_keyframes_%d" % ori] = np.concatenate([[n], o])
        if hasattr(L, "mw_search_by_bow_frame_many"):        # relocalisation: several candidates against the frame at once
            cand = np.array([kA, kB, kA], np.int32); oo = i32(len(cand) * len(kb)); cc = i32(len(cand))
            assert L.mw_search_by_bow_frame_many(drv.w, len(cand), _p(cand), fb, len(kb), _p(oo), _p(cc), C.c_float(0.75), ori) == 0
            out["bow_frame_many_%d" % ori] = np.concatenate([cc, oo])

    # ---- SearchForInitialization ---------------------------------------------------------------------------------
    fi1, ki1, di1, ui1, pti1, _ = make(False, poses[0], 0.0, 0.0, px_noise=0.5, level0_frac=0.6)
    fi2, ki2, di2, ui2, pti2, _ = make(False, (rot(0.002, -0.004, 0.01), np.array([0.05, 0.0, 0.02], np.float32)), 3.0, 0.0, px_noise=0.5, level0_frac=0.6)
    prev = np.stack([ki1["x"], ki1["y"]], 1).astype(np.float32).copy()
    for rnd in range(2):                                   # second round starts from the updated vbPrevMatched, as Tracking does
        m12 = i32(len(ki1))
        n = L.mw_search_for_initialization(drv.w, fi1, fi2, _p(prev), _p(m12), P(40 if dense else 100, [15, 40, 100, 160]), C.c_float(P(0.9, [0.7, 0.9, 1.0])), 1)
        out["init_%d" % rnd] = np.concatenate([[n], m12, prev.view(np.int32).ravel()])

    # ---- SearchForTriangulation -----------------------------------------------------------------------------------
    for only_stereo, coarse in ((0, 0), (1, 0), (0, 1)):
        pairs = i32(2 * len(kkA)); npairs = C.c_int()
        n = L.mw_search_for_triangulation(drv.w, kA, kB, only_stereo, coarse, _p(pairs), len(kkA), C.byref(npairs), C.c_float(0.6), 1)
        out["triang_%d_%d" % (only_stereo, coarse)] = np.concatenate([[n, npairs.value], pairs])
    pairs = i32(2 * len(kkA)); npairs = C.c_int()
    n = L.mw_search_for_triangulation(drv.w, kA, kB, 0, 0, _p(pairs), len(kkA), C.byref(npairs), C.c_float(0.6), 0)
    out["triang_noori"] = np.concatenate([[n, npairs.value], pairs])
    if hasattr(L, "mw_implicit_cache_probe"):                # the facade's implicit resident cache must notice a key frame whose content changed in place
        out["implicit_cache_probe"] = np.array([L.mw_implicit_cache_probe(drv.w, kA, kB)], np.int32)
    if hasattr(L, "mw_implicit_cache_lru_probe"):            # ... and is bounded: least recently used out, EraseImplicit releases at once
        out["implicit_cache_lru_probe"] = np.array([L.mw_implicit_cache_lru_probe(drv.w, kA, kB)], np.int32)

    # the same against several neighbours: the facade answers them in one call over device-resident key frames (twice: the second round
    # reuses the cached uploads), the reference is called once per neighbour
    if hasattr(L, "mw_search_for_triangulation_neighbours"):
        neigh = np.array([kB, kB, kB], np.int32); cap = len(kkA)
        for only_stereo, coarse, ori in ((0, 0, 1), (1, 1, 0)):
            pairs = i32(2 * cap * len(neigh)); npairs = i32(len(neigh)); nms = i32(len(neigh))
            rc = L.mw_search_for_triangulation_neighbours(drv.w, kA, len(neigh), _p(neigh), only_stereo, coarse, _p(pairs), cap, _p(npairs), _p(nms), C.c_float(0.6), ori, 2)
            assert rc == 0
            out["triang_neighbours_%d_%d" % (only_stereo, coarse)] = np.concatenate([npairs, nms, pairs])

    # ---- SearchBySim3 -----------------------------------------------------------------------------------------------
    R1, t1 = poses[0]; R2, t2 = poses[1]                    # S12 = T1w * Tw2 (scale 1) with a small perturbation
    R12 = (R1 @ R2.T).astype(np.float32); t12 = (t1 - R12 @ t2 + np.array([0.01, -0.01, 0.02])).astype(np.float32)
    for s in (1.0, 0.97):
        m12 = np.where(rng.uniform(size=len(kkA)) < 0.1, idsA, -1).astype(np.int32)
        n = L.mw_search_by_sim3(drv.w, kA, kB, _p(m12), C.c_float(s), _p(R12), _p(t12), C.c_float(P(7.5, [1.5, 4.0, 7.5, 15.0])))
        out["sim3_%d" % int(s * 100)] = np.concatenate([[n], m12])

    # ---- Fuse ----------------------------------------------------------------------------------------------------------
    kF, kkF, dF, uF, ptF, idsF = make(True, poses[2], 0.0, 0.45)
    drv.set_map_points(True, kF, idsF)
    # candidates: other map points of the same scene points (duplicates to be fused) + unrelated ones
    dup = []
    for i in rng.permutation(len(sc.X))[:500]:
        PO = sc.X[i]; d = float(np.linalg.norm(PO)); maxd = d * SCALE ** int(sc.level0[i])
        dup.append(drv.mappoint(sc.X[i] + rng.normal(0, 0.01, 3), PO / d, maxd / SCALE ** (NLEVELS - 1), maxd, sc.noisy_desc(np.array([i]), 25)[0], bad=rng.uniform() < 0.03,
                                n_obs=int(rng.integers(0, 6))))
    cand = np.concatenate([np.array(dup), idsF[idsF >= 0][:40], [-1, -1]]).astype(np.int32)
    rng.shuffle(cand)
    n = L.mw_fuse(drv.w, kF, _p(cand), len(cand), C.c_float(P(3.0, [0.7, 1.5, 2.5, 3.0, 6.0])), 0)
    st = np.array([drv.mappoint_state(int(c), kF) for c in cand if c >= 0]).ravel()
    st_kf = np.array([drv.mappoint_state(int(c), kF) for c in idsF if c >= 0]).ravel()
    out["fuse"] = np.concatenate([[n], drv.get_map_points(True, kF, len(kkF)), st, st_kf])

    # the same with a search radius below the chi-square bound (th = 1: radius 1 x scale, the test admits |e| up to 2.8 x scale): Fuse has no radius test on
    # the right coordinate, only the chi-square one (round 5: the product applied SearchByProjection's right-coordinate gate here as well; invisible at the
    # reference's th = 3, found by tools/soak_search_fuzz.py)
    kF1, kkF1, dF1, uF1, ptF1, idsF1 = make(True, poses[2], 0.0, 0.45)
    drv.set_map_points(True, kF1, idsF1)
    dup1 = []
    for i in rng.permutation(len(sc.X))[:500]:
        PO = sc.X[i]; d = float(np.linalg.norm(PO)); maxd = d * SCALE ** int(sc.level0[i])
        # moved along the viewing ray of this key frame: (u, v) stay, the right coordinate moves by 1.2 - 2.6 pixels of the predicted level
        Ow1 = -poses[2][0].T.astype(np.float64) @ poses[2][1].astype(np.float64)
        ray = sc.X[i] - Ow1; z = float(np.linalg.norm(ray))
        eps = float(rng.uniform(1.2, 2.6)) * SCALE ** int(sc.level0[i]) * z / BF * float(rng.choice([-1.0, 1.0]))
        dup1.append(drv.mappoint(Ow1 + ray * (1.0 + eps) + rng.normal(0, 0.002, 3), PO / d, maxd / SCALE ** (NLEVELS - 1), maxd * 1.3,
                                 sc.noisy_desc(np.array([i]), 25)[0], n_obs=int(rng.integers(0, 6))))
    cand1 = np.array(dup1, np.int32)
    n = L.mw_fuse(drv.w, kF1, _p(cand1), len(cand1), C.c_float(1.0), 0)
    st1 = np.array([drv.mappoint_state(int(c), kF1) for c in cand1]).ravel()
    out["fuse_th1"] = np.concatenate([[n], drv.get_map_points(True, kF1, len(kkF1)), st1])

    kG, kkG, dG, uG, ptG, idsG = make(True, poses[1], 0.0, 0.45)
    drv.set_map_points(True, kG, idsG)
    cand = np.array(dup, np.int32)[rng.permutation(len(dup))[:400]]
    rep = i32(len(cand))
    n = L.mw_fuse_sim3(drv.w, kG, C.c_float(1.03), _p(Rs), _p((ts * 1.03).astype(np.float32)), _p(cand), len(cand), C.c_float(P(4.0, [1.0, 2.0, 4.0, 9.0])), _p(rep))
    st = np.array([drv.mappoint_state(int(c), kG) for c in cand]).ravel()
    out["fuse_sim3"] = np.concatenate([[n], rep, drv.get_map_points(True, kG, len(kkG)), st])

    # ---- DescriptorDistance ---------------------------------------------------------------------------------------
    a = rng.integers(0, 256, (64, 32), dtype=np.uint8); b = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    b[:8] = a[:8]; b[8:16] = ~a[8:16]
    out["distance"] = np.array([L.mw_descriptor_distance(_p(a[i]), _p(b[i])) for i in range(64)], np.int32)
    return out


def build_and_run_rig(drv, seed):
    """The fisheye-rig (Nleft != -1) branches of the two SearchByProjection overloads."""
    rng = np.random.default_rng(seed)
    out = {}
    L = drv.L
    sc = Scene(rng)
    cam = drv.camera(); cam2 = drv.camera(FX * 1.01, FY * 0.99, CX + 3, CY - 2)
    mp_ids = []
    for i in range(len(sc.X)):
        d = float(np.linalg.norm(sc.X[i])); maxd = d * SCALE ** int(sc.level0[i])
        mp_ids.append(drv.mappoint(sc.X[i], sc.X[i] / d, maxd / SCALE ** (NLEVELS - 1), maxd, sc.noisy_desc(np.array([i]), 20)[0], bad=rng.uniform() < 0.04, n_obs=int(rng.integers(0, 6))))
    mp_ids = np.array(mp_ids)
    trl = (rot(0.0, 0.02, 0.0), np.array([-0.11, 0.0, 0.0], np.float32))

    def make_rig(pose, rot_off, assoc_frac, keyframe=False):
        R, t = pose
        kl, dl, _, ptl, nodel = sc.observe(R, t, clutter=150, rot_off=rot_off)
        Rr = (trl[0] @ R).astype(np.float32); tr = (trl[0] @ t + trl[1]).astype(np.float32)
        kr, dr, _, ptr, noder = sc.observe(Rr, tr, clutter=150, rot_off=rot_off)
        fid = drv.frame(keyframe, kl, np.concatenate([dl, dr]), None, R, t, cam, cam2, keys_right=kr, trl=trl)
        drv.set_feat_vec(keyframe, fid, np.concatenate([nodel, noder]))
        if keyframe:
            pt = np.concatenate([ptl, ptr])
            ids = np.where((pt >= 0) & (rng.uniform(size=len(pt)) < assoc_frac), mp_ids[np.maximum(pt, 0)], -1).astype(np.int32)
            return fid, len(kl), len(kr), ids, (R, t, Rr, tr)
        pt = np.concatenate([ptl, ptr])
        ids = np.where((pt >= 0) & (rng.uniform(size=len(pt)) < assoc_frac), mp_ids[np.maximum(pt, 0)], -1).astype(np.int32)
        # stereo correspondences between the two cameras of the rig
        l2r = np.full(len(kl), -1, np.int32); r2l = np.full(len(kr), -1, np.int32)
        where_r = {p: j for j, p in enumerate(ptr) if p >= 0}
        for i, p in enumerate(ptl):
            if p >= 0 and p in where_r and rng.uniform() < 0.7:
                l2r[i] = where_r[p]; r2l[where_r[p]] = i
        L.mw_set_lr_matches(drv.w, fid, _p(l2r), _p(r2l))
        return fid, len(kl), len(kr), ids, (R, t, Rr, tr)

    pose = (rot(0.01, -0.02, 0.03), np.array([0.25, -0.03, 0.35], np.float32))
    f0, nl, nr, ids0, (R, t, Rr, tr) = make_rig(pose, 0.0, 0.3)
    Xc = (R @ sc.X.T).T + t; Xr = (Rr @ sc.X.T).T + tr
    for i in range(len(sc.X)):
        z = Xc[i, 2]; zr = Xr[i, 2]
        uu = FX * Xc[i, 0] / z + CX; vv = FY * Xc[i, 1] / z + CY
        ur = FX * 1.01 * Xr[i, 0] / zr + CX + 3; vr = FY * 0.99 * Xr[i, 1] / zr + CY - 2
        inv = z > 0.5 and 0 <= uu < W and 0 <= vv < H and rng.uniform() < 0.85
        invr = zr > 0.5 and 0 <= ur < W and 0 <= vr < H and rng.uniform() < 0.85
        lvl = int(np.clip(sc.level0[i] + rng.integers(-1, 2), 0, NLEVELS - 1))
        lvlr = int(np.clip(sc.level0[i] + rng.integers(-1, 2), 0, NLEVELS - 1)) if rng.uniform() < 0.9 else -1
        drv.track(mp_ids[i], inv, invr, lvl, lvlr, [uu, vv, ur, rng.choice([0.9985, 0.99]), z, vr, rng.choice([0.9985, 0.99])])
    sel = rng.permutation(len(mp_ids))[:700]
    for th in (1.0, 3.0):
        drv.set_map_points(False, f0, ids0)
        n = L.mw_search_by_projection_mappoints(drv.w, f0, _p(mp_ids[sel].astype(np.int32)), len(sel), C.c_float(P(th, [0.7, 1.5, 3.0, 6.0, 12.0])), 0, C.c_float(50.0), C.c_float(P(0.8, [0.6, 0.8, 0.95])), 1)
        out["rig_sbp_mappoints_%d" % int(th)] = np.concatenate([[n], drv.get_map_points(False, f0, nl + nr)])

    for tag, pose_c in (("fwd", pose), ("near", (rot(0.0, 0.0, 0.01), np.array([0.01, 0.0, 0.02], np.float32)))):
        fl, nll, nrl, idsl, _ = make_rig((np.eye(3, dtype=np.float32), np.zeros(3, np.float32)), 0.0, 0.8)
        drv.set_map_points(False, fl, idsl, outlier=(rng.uniform(size=nll + nrl) < 0.05))
        fc, nlc, nrc, idsc, _ = make_rig(pose_c, 6.0, 0.1)
        for ori in (1, 0):
            drv.set_map_points(False, fc, idsc)
            n = L.mw_search_by_projection_frame(drv.w, fc, fl, C.c_float(P(10.0, [1.5, 4.0, 10.0, 25.0])), 0, C.c_float(P(0.9, [0.7, 0.9, 1.0])), ori)
            out["rig_sbp_frame_%s_%d" % (tag, ori)] = np.concatenate([[n], drv.get_map_points(False, fc, nlc + nrc)])

    # ---- Fuse(pKF, vpMapPoints, th, bRight) on a rig key frame: left camera, then right camera (LocalMapping::SearchInNeighbors) ----
    kF, nlF, nrF, idsF, _ = make_rig(pose, 0.0, 0.4, keyframe=True)
    drv.set_map_points(True, kF, idsF)
    dup = []
    for i in rng.permutation(len(sc.X))[:500]:
        d = float(np.linalg.norm(sc.X[i])); maxd = d * SCALE ** int(sc.level0[i])
        dup.append(drv.mappoint(sc.X[i] + rng.normal(0, 0.01, 3), sc.X[i] / d, maxd / SCALE ** (NLEVELS - 1), maxd, sc.noisy_desc(np.array([i]), 25)[0], bad=rng.uniform() < 0.03,
                                n_obs=int(rng.integers(0, 6))))
    cand = np.concatenate([np.array(dup), idsF[idsF >= 0][:30], [-1]]).astype(np.int32)
    rng.shuffle(cand)
    for right in (0, 1):
        n = L.mw_fuse(drv.w, kF, _p(cand), len(cand), C.c_float(P(3.0, [0.7, 1.5, 3.0, 6.0])), right)
        st = np.array([drv.mappoint_state(int(c), kF) for c in cand if c >= 0]).ravel()
        out["rig_fuse_%d" % right] = np.concatenate([[n], drv.get_map_points(True, kF, nlF + nrF), st])

    # ---- SearchByBoW on rig objects: key frame -> rig frame (both cameras), key frame -> key frame (camera 1 only) ----
    kA, nlA, nrA, idsA, _ = make_rig((np.eye(3, dtype=np.float32), np.zeros(3, np.float32)), 0.0, 0.7, keyframe=True)
    drv.set_map_points(True, kA, idsA)
    fb, nlb, nrb, _, _ = make_rig(pose, 15.0, 0.0)
    for ori in (1, 0):
        o = np.full(nlb + nrb, -1, np.int32)
        n = L.mw_search_by_bow_frame(drv.w, kA, fb, _p(o), C.c_float(P(0.75, [0.55, 0.75, 0.95])), ori)
        out["rig_bow_frame_%d" % ori] = np.concatenate([[n], o])
        o = np.full(nlA + nrA, -1, np.int32)
        n = L.mw_search_by_bow_keyframes(drv.w, kA, kF, _p(o), C.c_float(P(0.8, [0.55, 0.8, 0.95])), ori)
        out["rig_bow_keyframes_%d" % ori] = np.concatenate([[n], o])
    return out


def kb8_project(p8, Xc):
    """KannalaBrandt8::project (src/CameraModels/KannalaBrandt8.cpp:87-104) in fp64, for building scenes"""
    x, y, z = Xc[:, 0], Xc[:, 1], Xc[:, 2]
    theta = np.arctan2(np.sqrt(x * x + y * y), z); psi = np.arctan2(y, x)
    r = theta + p8[4] * theta ** 3 + p8[5] * theta ** 5 + p8[6] * theta ** 7 + p8[7] * theta ** 9
    return p8[0] * r * np.cos(psi) + p8[2], p8[1] * r * np.sin(psi) + p8[3]


def build_and_run_kb8(drv, seed):
    """ORBmatcher::SearchForTriangulation on key frames with Kannala-Brandt cameras (src/ORBmatcher.cc:1045-1323, the branches of :1067-1083,
    :1126-1141, :1189, :1203-1240): one fisheye camera per key frame, and the two-camera rig (features of both cameras in one key frame)."""
    rng = np.random.default_rng(seed)
    out = {}
    L = drv.L
    sc = Scene(rng)
    # a 640 x 480 fisheye pair in the style of Examples/Stereo/TUM-VI.yaml, scaled to the world's image size
    P1 = np.array([FX * 0.62, FY * 0.62, CX + 1.5, CY - 2.0, 0.0035, 0.0007, -0.0020, 0.0002], np.float32)
    P2 = np.array([FX * 0.618, FY * 0.621, CX - 2.0, CY + 1.0, 0.0034, 0.0018, -0.0027, 0.0003], np.float32)
    trl = (rot(0.0, 0.03, 0.001), np.array([-0.101, 0.0006, 0.001], np.float32))
    if FUZZ is not None:                 # variant "kb8fuzz": random focal lengths, principal points, distortion coefficients, rig rotation and baseline
        f = float(FUZZ.uniform(0.45, 0.9))
        P1 = np.array([FX * f, FY * f * float(FUZZ.uniform(0.98, 1.02)), CX + float(FUZZ.uniform(-12, 12)), CY + float(FUZZ.uniform(-12, 12))] +
                      [float(FUZZ.normal(0, sd)) for sd in (0.02, 0.008, 0.004, 0.001)], np.float32)
        P2 = np.array([P1[0] * float(FUZZ.uniform(0.98, 1.02)), P1[1] * float(FUZZ.uniform(0.98, 1.02)), CX + float(FUZZ.uniform(-12, 12)), CY + float(FUZZ.uniform(-12, 12))] +
                      [float(FUZZ.normal(0, sd)) for sd in (0.02, 0.008, 0.004, 0.001)], np.float32)
        trl = (rot(*FUZZ.normal(0, 0.02, 3)), np.array([-float(FUZZ.uniform(0.05, 0.25)), float(FUZZ.normal(0, 0.002)), float(FUZZ.normal(0, 0.002))], np.float32))
    L.mw_add_camera_kb8.restype = C.c_int
    c1 = L.mw_add_camera_kb8(drv.w, _p(P1)); c2 = L.mw_add_camera_kb8(drv.w, _p(P2))
    mp_ids = []
    for i in range(len(sc.X)):
        d = float(np.linalg.norm(sc.X[i])); maxd = d * SCALE ** int(sc.level0[i])
        mp_ids.append(drv.mappoint(sc.X[i], sc.X[i] / d, maxd / SCALE ** (NLEVELS - 1), maxd, sc.noisy_desc(np.array([i]), 20)[0], bad=0, n_obs=2))
    mp_ids = np.array(mp_ids)

    def make(pose, rig, assoc_frac):
        R, t = pose
        kl, dl, url, ptl, nodel = sc.observe(R, t, clutter=120, px_noise=0.6, maxflips=35, proj=lambda X: kb8_project(P1.astype(np.float64), X))
        if not rig:
            kid = drv.frame(True, kl, dl, None, R, t, c1)
            drv.set_feat_vec(True, kid, nodel)
            pt = ptl
        else:
            Rr = (trl[0] @ R).astype(np.float32); tr = (trl[0] @ t + trl[1]).astype(np.float32)
            kr, dr, _, ptr, noder = sc.observe(Rr, tr, clutter=120, px_noise=0.6, maxflips=35, proj=lambda X: kb8_project(P2.astype(np.float64), X))
            kid = drv.frame(True, kl, np.concatenate([dl, dr]), None, R, t, c1, c2, keys_right=kr, trl=trl)
            drv.set_feat_vec(True, kid, np.concatenate([nodel, noder]))
            pt = np.concatenate([ptl, ptr])
        ids = np.where((pt >= 0) & (rng.uniform(size=len(pt)) < assoc_frac), mp_ids[np.maximum(pt, 0)], -1).astype(np.int32)
        drv.set_map_points(True, kid, ids)
        return kid, len(pt)

    poseA = (np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
    poseB = (rot(0.01, -0.03, 0.02), np.array([0.45, -0.05, 0.12], np.float32))
    for rig in (0, 1):
        kA, nA = make(poseA, rig, 0.3); kB, nB = make(poseB, rig, 0.3)
        for (only_stereo, coarse, ori) in ((0, 0, 1), (0, 0, 0), (0, 1, 1), (1, 0, 0)):
            pairs = np.full((max(nA, 1), 2), -1, np.int32); npairs = C.c_int(0)
            n = L.mw_search_for_triangulation(drv.w, kA, kB, only_stereo, coarse, _p(pairs), len(pairs), C.byref(npairs), C.c_float(0.6), ori)
            out["kb8_triangulation_rig%d_%d%d%d" % (rig, only_stereo, coarse, ori)] = np.concatenate([[n, npairs.value], pairs[:npairs.value].ravel()])
        # several neighbours at once: the facade's one call over device-resident key frames (its Kannala-Brandt kernel), the reference once per neighbour
        if hasattr(L, "mw_search_for_triangulation_neighbours"):
            kC, nC = make((rot(-0.02, 0.02, 0.01), np.array([-0.3, 0.04, 0.08], np.float32)), rig, 0.3)
            neigh = np.array([kB, kC, kB], np.int32); cap = max(nA, 1)
            for (only_stereo, coarse, ori) in ((0, 0, 1), (1, 1, 0)):
                pairs = np.full(2 * cap * len(neigh), -1, np.int32); npairs = np.full(len(neigh), -1, np.int32); nms = np.full(len(neigh), -1, np.int32)
                rc = L.mw_search_for_triangulation_neighbours(drv.w, kA, len(neigh), _p(neigh), only_stereo, coarse, _p(pairs), cap, _p(npairs), _p(nms), C.c_float(0.6), ori, 2)
                assert rc == 0
                out["kb8_triangulation_neighbours_rig%d_%d%d%d" % (rig, only_stereo, coarse, ori)] = np.concatenate([npairs, nms, pairs])
    return out


if __name__ == "__main__":
    drv_path, orbx_path, seed, variant, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    d = Driver(drv_path, orbx_path or None)
    if variant in ("fuzz", "rigfuzz", "kb8fuzz"):
        FUZZ = np.random.default_rng(50000 + seed); variant = {"fuzz": "base", "rigfuzz": "rig", "kb8fuzz": "kb8"}[variant]
    res = build_and_run_rig(d, seed) if variant == "rig" else build_and_run_kb8(d, seed) if variant == "kb8" else build_and_run(d, seed, variant)
    extra = {}
    d.close()
    if len(sys.argv) > 6 and sys.argv[6] == "twice" and orbx_path:
        # tests/test_lifetime.py: the same world built, searched and destroyed again must leave the library holding what it held after the first
        # (the facade's per-thread handle and its scratch stay; key frames, caches and everything per world have to be gone)
        def live():
            a = (C.c_longlong * 4)(); assert d._orbx.orbx_debug_live_resources(a) == 0; return list(a)
        first = live()
        for _ in range(int(sys.argv[7]) if len(sys.argv) > 7 else 1):
            d2 = Driver(drv_path, None)
            res2 = build_and_run_rig(d2, seed) if variant == "rig" else build_and_run_kb8(d2, seed) if variant == "kb8" else build_and_run(d2, seed, variant)
            assert all(np.array_equal(res[k], res2[k]) for k in res)
            d2.close()
        extra = dict(live_first=np.array(first), live_again=np.array(live()))
    np.savez(dst, flavour=np.frombuffer(d.L.mw_flavour(), np.uint8), **res, **extra)
