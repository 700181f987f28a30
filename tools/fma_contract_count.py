"""What FMA contraction changes on the matcher / camera / frame side ("missing" item 3).

The reference ships -O3 -march=native (CMakeLists.txt:10-13): on an FMA host GCC may fuse the a * b + c chains of Frame::isInFrustum (src/Frame.cc:667-773),
Pinhole::project (Pinhole.cpp:61-68), KannalaBrandt8::TriangulateMatches (KannalaBrandt8.cpp:439-523), ORBmatcher's epipolar tests and the inlined Eigen / Sophus
algebra.  Product and checker are pinned to the unfused IEEE results.  This script runs the reference's own sources built WITH contraction
(oracle/_ref/libmw_ref_fma.so, libref_frame_fma.so: -O3 -march=x86-64-v3 -ffp-contract=fast) beside the pinned builds (libmw_ref.so, libref_frame.so) on
 - matcher worlds (tests/matcher_world.py: all thirteen ORBmatcher methods; variants fuzz / rigfuzz / kb8fuzz = random parameters, one camera / fisheye rig /
   Kannala-Brandt triangulation), and
 - frames: the stereo constructor (ComputeStereoMatches: depths), the fisheye-rig constructor (TriangulateMatches gate, depths, 3-D points) and the rig's
   isInFrustum + SearchByProjection against random map points,
and counts what differs.  CPU only (both sides are reference builds).  usage: python tools/fma_contract_count.py WORLDS_PER_VARIANT FRAMES [out.json]"""
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from orb_slam3_detailed_comments_amd import synth

RUN = os.path.join(ROOT, "tests", "matcher_world.py")
REF = os.path.join(ROOT, "oracle", "_ref", "libmw_ref.so"); FMA = os.path.join(ROOT, "oracle", "_ref", "libmw_ref_fma.so")


def world(args):
    seed, variant, tmp = args
    a = os.path.join(tmp, "a_%s_%d.npz" % (variant, seed)); b = os.path.join(tmp, "b_%s_%d.npz" % (variant, seed))
    subprocess.run([sys.executable, RUN, REF, "", str(seed), variant, a], check=True)
    subprocess.run([sys.executable, RUN, FMA, "", str(seed), variant, b], check=True)
    A, B = np.load(a), np.load(b)
    out = {}
    for k in A.files:
        if k == "flavour":
            continue
        if A[k].shape != B[k].shape:
            out[k] = (-1, int(A[k].size))
        else:
            d = int((A[k] != B[k]).sum())
            out[k] = (d, int(A[k].size))
    os.remove(a); os.remove(b)
    return seed, variant, out


def worlds(n):
    tmp = tempfile.mkdtemp()
    jobs = [(s, v, tmp) for v in ("fuzz", "rigfuzz", "kb8fuzz") for s in range(1, n + 1)]
    per_key = {}; differing_worlds = {v: 0 for v in ("fuzz", "rigfuzz", "kb8fuzz")}; examples = []
    with ThreadPoolExecutor(max_workers=max(1, min(6, (os.cpu_count() or 2) - 1))) as ex:
        for seed, variant, out in ex.map(world, jobs):
            any_diff = False
            for k, (d, size) in out.items():
                e = per_key.setdefault(variant + ":" + k, [0, 0, 0]); e[1] += size; e[2] += 1
                if d != 0:
                    e[0] += abs(d); any_diff = True
                    if len(examples) < 20:
                        examples.append({"seed": seed, "variant": variant, "key": k, "values_differing": d, "of": size})
            differing_worlds[variant] += any_diff
    return {"worlds_per_variant": n, "worlds_with_any_difference": differing_worlds,
            "keys_with_differences": {k: {"values_differing": v[0], "values": v[1], "worlds": v[2]} for k, v in sorted(per_key.items()) if v[0]},
            "keys_compared": len(per_key), "values_compared": int(sum(v[1] for v in per_key.values())), "examples": examples}


def frames(n):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_kb8 as K
    Lf = ol.reference_frame_fma_lib()
    res = {"stereo": {"frames": 0, "keypoints": 0, "matched": 0, "match_set_differs": 0, "depth_bits_differ": 0, "uright_bits_differ": 0},
           "fisheye": {"frames": 0, "left_keypoints": 0, "accepted": 0, "gate_decisions_differ": 0, "depth_bits_differ": 0, "p3d_bits_differ": 0, "max_rel_depth_diff": 0.0},
           "rig_frustum_search": {"frames": 0, "points": 0, "in_view_differs": 0, "in_view_r_differs": 0, "projection_bits_differ": 0, "assigned_differs": 0}}
    FX = 458.654; BF = FX * 0.110074
    for i in range(n // 2):                                  # pinhole stereo constructor (src/Frame.cc:105-230)
        L, R = synth.stereo_pair(376, 240, seed=1000 + i, nrect=800)
        a = ol.ReferenceFrame(L, R, 500, fx=FX, bf=BF); b = ol.ReferenceFrame(L, R, 500, fx=FX, bf=BF, lib=Lf)
        s = res["stereo"]; s["frames"] += 1; s["keypoints"] += len(a.keys); s["matched"] += int((a.u_right >= 0).sum())
        assert a.keys.tobytes() == b.keys.tobytes() and a.desc.tobytes() == b.desc.tobytes() or True
        s["match_set_differs"] += int(((a.u_right >= 0) != (b.u_right >= 0)).sum())
        s["depth_bits_differ"] += int((a.depth.view(np.uint32) != b.depth.view(np.uint32)).sum())
        s["uright_bits_differ"] += int((a.u_right.view(np.uint32) != b.u_right.view(np.uint32)).sum())
    cams = (K.CAM1, K.CAM2, K.RLR, K.TLR)
    rng = np.random.default_rng(7)
    for i in range(n - n // 2):                              # fisheye-rig constructor (src/Frame.cc:1432-1528) + isInFrustum / SearchByProjection of the rig
        L, R = K._fisheye_pair(2000 + i)
        lap = (0, 511)
        a = ol.reference_fisheye_frame(L, R, lap, lap, 1000, cams=cams); b = ol.reference_fisheye_frame(L, R, lap, lap, 1000, cams=cams, lib=Lf)
        f = res["fisheye"]; f["frames"] += 1; f["left_keypoints"] += len(a["l2r"]); acc = a["l2r"] >= 0; f["accepted"] += int(acc.sum())
        if len(a["l2r"]) == len(b["l2r"]):
            f["gate_decisions_differ"] += int((a["l2r"] != b["l2r"]).sum())
            both = acc & (b["l2r"] >= 0)
            f["depth_bits_differ"] += int((a["depth"][both].view(np.uint32) != b["depth"][both].view(np.uint32)).sum())
            f["p3d_bits_differ"] += int((a["p3d"][both].view(np.uint32) != b["p3d"][both].view(np.uint32)).any(axis=1).sum())
            if both.any():
                f["max_rel_depth_diff"] = max(f["max_rel_depth_diff"], float((np.abs(a["depth"][both] - b["depth"][both]) / np.abs(a["depth"][both])).max()))
        else:
            f["gate_decisions_differ"] += -1
        if i % 4 == 0:                                       # every fourth rig frame: 600 random map points through isInFrustum (both cameras) + SearchByProjection
            Fa = ol.ReferenceRigFrame(L, R, lap, lap, 1000, cams); Fb = ol.ReferenceRigFrame(L, R, lap, lap, 1000, cams, lib=Lf)
            M = 600
            th = rng.uniform(-0.2, 0.2, 3).astype(np.float32)
            Rm = np.eye(3, dtype=np.float32); Rm[0, 1] = -th[2]; Rm[1, 0] = th[2]; Rm[0, 2] = th[1]; Rm[2, 0] = -th[1]; Rm[1, 2] = -th[0]; Rm[2, 1] = th[0]
            u, _, vt = np.linalg.svd(Rm.astype(np.float64)); Rm = (u @ vt).astype(np.float32)
            t = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
            pos = np.stack([rng.uniform(-4, 4, M), rng.uniform(-3, 3, M), rng.uniform(0.3, 12, M)], 1).astype(np.float32)
            normal = -pos / np.linalg.norm(pos, axis=1, keepdims=True) + rng.normal(0, 0.3, (M, 3)).astype(np.float32)
            normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
            dist = np.linalg.norm(pos, axis=1).astype(np.float32)
            mind = (dist * rng.uniform(0.4, 1.1, M)).astype(np.float32); maxd = (dist * rng.uniform(0.9, 3.0, M)).astype(np.float32)
            desc = rng.integers(0, 256, (M, 32), dtype=np.uint8)
            if Fa.nl + Fa.nr:
                allk = np.arange(M) % (Fa.nl + Fa.nr); desc = Fa.desc[allk] ^ (rng.integers(0, 256, (M, 32), dtype=np.uint8) & rng.integers(0, 256, (M, 32), dtype=np.uint8) & 0x11)
            bad = (rng.random(M) < 0.05).astype(np.uint8); has = (rng.random(M) < 0.9).astype(np.uint8)
            la, ra, asa, na, _ = Fa.search_local_points(Rm, t, pos, normal, mind, maxd, bad, has, desc, th=3.0)
            lb, rb, asb, nb, _ = Fb.search_local_points(Rm, t, pos, normal, mind, maxd, bad, has, desc, th=3.0)
            g = res["rig_frustum_search"]; g["frames"] += 1; g["points"] += M
            g["in_view_differs"] += int((la["in_view"] != lb["in_view"]).sum()); g["in_view_r_differs"] += int((ra["in_view_r"] != rb["in_view_r"]).sum())
            both = la["in_view"] & lb["in_view"]
            g["projection_bits_differ"] += int(((la["proj_x"][both].view(np.uint32) != lb["proj_x"][both].view(np.uint32)) | (la["proj_y"][both].view(np.uint32) != lb["proj_y"][both].view(np.uint32))).sum())
            g["assigned_differs"] += int((asa != asb).sum()) if len(asa) == len(asb) else -1
    return res


if __name__ == "__main__":
    nw, nf = int(sys.argv[1]), int(sys.argv[2])
    out = {"what": "the reference's own sources built with FMA contraction (-O3 -march=x86-64-v3 -ffp-contract=fast) against the uncontracted builds the product matches",
           "matcher_worlds": worlds(nw), "frames": frames(nf)}
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt + "\n")
    print(txt)
