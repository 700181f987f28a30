"""Measurement of the "next" rows of SURVEY.md §8f and the guided searches at BASELINE.json config-4 sizes: wall time of the device path
(through the C ABI, host views in / results out, i.e. including uploads, the ordered host replay and downloads) next to the CPU
oracle / reference on the same inputs.  Not the headline benchmark (bench.py); run by tools/gpu_round2.sh, results go to profiles/."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
import search_scenes as sc
import vocab_scenes as vs
from orb_slam3_detailed_comments_amd import synth, ORBextractor
from orb_slam3_detailed_comments_amd import matcher as M
from orb_slam3_detailed_comments_amd.vocabulary import ORBVocabulary


def best(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


def main():
    out = {}
    rng = np.random.default_rng(0)
    w, h, nf, MP = 640, 480, 1000, 5000                       # BASELINE.json configs[3]: TUM RGB-D, 5000 local map points
    ex = ORBextractor(nf, 1.2, 8, 20, 7)
    img = synth.corner_field(w, h, seed=300, nrect=int(3000 * w * h / (752 * 480)))
    fv, k, d, u, scales = sc.frame_from_image(img, nf, rng)
    mps = sc.map_points_for_frame(k, d, u, scales, MP, rng, w, h)
    last = sc.last_frame_for(k, d, scales, rng, w, h, 40.0)
    pts = sc.projected_points(k, d, u, scales, rng, w, h, M=MP)
    m = M.ORBmatcher(0.8)
    rows = [
        ("SearchByProjection(Frame, 5000 MapPoints)", lambda: m.SearchByProjection(ex, fv, mps, 3.0), lambda: ol.oracle_search_by_projection_mappoints(fv, mps, 3.0, False, 0.0, 0.8)),
        ("SearchByProjection(Frame, LastFrame)", lambda: m.SearchByProjectionFrame(ex, fv, last, 7.0), lambda: ol.oracle_search_by_projection_frame(fv, last, 7.0, False, False, True)),
        ("SearchByProjection(KeyFrame, Sim3, 5000 points)", lambda: m.SearchByProjectionSim3(ex, fv, pts, 8, 1.0), lambda: ol.oracle_search_by_projection_sim3(fv, pts, 8, 1.0)),
        ("Fuse candidate search (5000 points, chi2 gate)", lambda: m.FuseCandidates(ex, fv, pts, 3.0, (1.0 / np.asarray(scales) ** 2).astype(np.float32)),
         lambda: ol.oracle_fuse_candidates(fv, pts, 3.0, (1.0 / np.asarray(scales) ** 2).astype(np.float32))),
    ]
    (kf1, kf2), F12, ep = sc.keyframe_pair(rng, nf, w, h)
    rows += [
        ("SearchForTriangulation", lambda: M.ORBmatcher(0.6, False).SearchForTriangulation(ex, kf1[0], kf2[0], F12, ep), lambda: ol.oracle_search_for_triangulation(kf1[0], kf2[0], F12, ep, False, False, False)),
        ("SearchByBoW(KeyFrame, Frame)", lambda: M.ORBmatcher(0.7, True).SearchByBoW(ex, kf1[0], kf2[0], True), lambda: ol.oracle_search_by_bow(kf1[0], kf2[0], 0.7, True, True)),
    ]
    # the same against 20 neighbours in one launch vs 20 CPU calls
    neigh = [kf2[0]] * 20
    Fs = np.tile(np.asarray(F12, np.float32), (20, 1)); Es = np.tile(np.asarray(ep, np.float32), (20, 1))
    rows.append(("SearchForTriangulation x 20 neighbours (one batched launch)", lambda: M.ORBmatcher(0.6, False).SearchForTriangulationBatch(ex, kf1[0], neigh, Fs, Es),
                 lambda: [ol.oracle_search_for_triangulation(kf1[0], kf2[0], F12, ep, False, False, False) for _ in range(20)]))
    rows.append(("SearchByBoW x 10 relocalisation candidates (one batched launch)", lambda: M.ORBmatcher(0.7, True).SearchByBoWBatch(ex, [kf1[0]] * 10, [kf2[0]] * 10, True),
                 lambda: [ol.oracle_search_by_bow(kf1[0], kf2[0], 0.7, True, True) for _ in range(10)]))
    # the same two with device-resident key frames (orbm_keyframe): uploaded once, a call moves flags, poses and results only; the SearchByBoW
    # accept loop runs on the device
    rk1, rk2 = M.ResidentKeyFrame(ex, kf1[0]), M.ResidentKeyFrame(ex, kf2[0])
    f1, f2 = kf1[0].keep[8], kf2[0].keep[8]
    rows.append(("SearchForTriangulation x 20 neighbours, device-resident key frames", lambda: M.ORBmatcher(0.6, False).SearchForTriangulationResident(ex, rk1, f1, [rk2] * 20, [f2] * 20, Fs, Es),
                 lambda: [ol.oracle_search_for_triangulation(kf1[0], kf2[0], F12, ep, False, False, False) for _ in range(20)]))
    rows.append(("SearchByBoW x 10 relocalisation candidates, device-resident key frames", lambda: M.ORBmatcher(0.7, True).SearchByBoWResident(ex, [rk1] * 10, [f1] * 10, [rk2] * 10, [f2] * 10, True),
                 lambda: [ol.oracle_search_by_bow(kf1[0], kf2[0], 0.7, True, True) for _ in range(10)]))
    rows.append(("SearchByBoW(KeyFrame, Frame), device-resident key frames", lambda: M.ORBmatcher(0.7, True).SearchByBoWResident(ex, [rk1], [f1], [rk2], [f2], True),
                 lambda: ol.oracle_search_by_bow(kf1[0], kf2[0], 0.7, True, True)))
    # MapPoint::ComputeDistinctiveDescriptors for 5000 map points with ~20 observations each
    counts = rng.integers(5, 40, MP); start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    dd = rng.integers(0, 256, (int(start[-1]), 32), dtype=np.uint8)
    rows.append(("ComputeDistinctiveDescriptors (5000 map points, %d descriptors)" % len(dd), lambda: M.ComputeDistinctiveDescriptors(ex, dd, start), lambda: ol.oracle_distinctive_descriptors(dd, start)))
    for name, gpu, cpu in rows:
        out[name] = {"gpu_ms": round(best(gpu, 5), 3), "cpu_oracle_ms": round(best(cpu, 2), 3)}
    # The four batched rows again at the C ABI, without the Python wrappers (argument arrays prebuilt, results left in their int arrays): a wrapper
    # costs 20-600 us per call (the list of matched pairs of 20 neighbours is built in Python), as much as the searches themselves.
    import ctypes as C
    L = ex._lib.L; OL = ol.oracle()
    N1 = kf1[0].view.N
    m20 = np.full((20, N1), -1, np.int32); nm20 = np.zeros(20, np.int32); m10 = [np.full(N1, -1, np.int32) for _ in range(10)]; nm10 = np.zeros(10, np.int32)
    vp = lambda xs: (C.c_void_p * len(xs))(*xs)
    k2v = vp([C.cast(kf2[0].ref(), C.c_void_p)] * 20); k1v10 = vp([C.cast(kf1[0].ref(), C.c_void_p)] * 10); k2v10 = vp([C.cast(kf2[0].ref(), C.c_void_p)] * 10)
    r2v = vp([rk2._kf] * 20); r1v10 = vp([rk1._kf] * 10); r2v10 = vp([rk2._kf] * 10)
    f2v = vp([f2.ctypes.data] * 20); f1v10 = vp([f1.ctypes.data] * 10); f2v10 = vp([f2.ctypes.data] * 10); po10 = vp([m.ctypes.data for m in m10])
    F9 = np.ascontiguousarray(F12, np.float32).reshape(9); E2 = np.ascontiguousarray(ep, np.float32).reshape(2); mo = np.full(N1, -1, np.int32)
    OL.orbo_search_for_triangulation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    OL.orbo_search_by_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]

    def cpu_sft20():
        for _ in range(20): OL.orbo_search_for_triangulation(kf1[0].ref(), kf2[0].ref(), F9.ctypes.data, E2.ctypes.data, 0, 0, 0, mo.ctypes.data)

    def cpu_bow10():
        for _ in range(10): OL.orbo_search_by_bow(kf1[0].ref(), kf2[0].ref(), 0.7, 1, 1, mo.ctypes.data)
    raw = [
        ("SearchForTriangulation x 20 neighbours", lambda: L.orbm_search_for_triangulation_batch(ex._h, kf1[0].ref(), 20, k2v, Fs.ctypes.data, Es.ctypes.data, 0, 0, 0, m20.ctypes.data, nm20.ctypes.data),
         lambda: L.orbm_search_for_triangulation_resident(ex._h, rk1._kf, f1.ctypes.data, 20, r2v, f2v, Fs.ctypes.data, Es.ctypes.data, 0, 0, 0, m20.ctypes.data, nm20.ctypes.data), cpu_sft20),
        ("SearchByBoW x 10 relocalisation candidates", lambda: L.orbm_search_by_bow_batch(ex._h, 10, k1v10, k2v10, 0.7, 1, 1, po10, nm10.ctypes.data),
         lambda: L.orbm_search_by_bow_resident(ex._h, 10, r1v10, f1v10, r2v10, f2v10, 0.7, 1, 1, po10, nm10.ctypes.data), cpu_bow10),
    ]
    for name, host_views, resident, cpu in raw:
        out[name + " (C ABI, no Python wrapper)"] = {"gpu_ms_host_views": round(best(host_views, 10), 3), "gpu_ms_resident_key_frames": round(best(resident, 10), 3),
                                                       "cpu_oracle_ms": round(best(cpu, 3), 3)}
    # The reference's own single-call signatures at the C ABI, the way the facade serves an UNCHANGED call site since round 5 (ORBmatcher::ImplicitCache:
    # key frames stay on the device between calls): SearchForTriangulation(pKF1, pKF2, ...) = one resident search with one neighbour;
    # SearchByBoW(pKF, F, ...) = the frame uploaded as a transient object + one resident search + its release.  Beside them the floor of ANY synchronous
    # device call: 32 bytes up, one tiny kernel, 4 bytes down, host waits (orbm_hamming_matrix on one pair of descriptors).
    r2v1 = vp([rk2._kf]); f2v1 = vp([f2.ctypes.data]); r1v1 = vp([rk1._kf]); f1v1 = vp([f1.ctypes.data]); po1 = vp([m10[0].ctypes.data])
    one = np.zeros(1, np.int32); da = np.zeros(32, np.uint8); db = np.ones(32, np.uint8)

    def bow_one():
        rf = C.c_void_p()
        L.orbm_keyframe_create(ex._h, kf2[0].ref(), C.byref(rf))
        L.orbm_search_by_bow_resident(ex._h, 1, r1v1, f1v1, vp([rf]), None, 0.7, 1, 1, po1, nm10.ctypes.data)
        L.orbm_keyframe_destroy(rf)
    floor = round(best(lambda: L.orbm_hamming_matrix(ex._h, da.ctypes.data, 1, db.ctypes.data, 1, one.ctypes.data), 50), 4)
    out["single calls at the C ABI, key frames resident (the unchanged facade call sites)"] = {
        "synchronous_round_trip_floor_ms": floor,
        "SearchForTriangulation(pKF1, pKF2) gpu_ms": round(best(lambda: L.orbm_search_for_triangulation_resident(ex._h, rk1._kf, f1.ctypes.data, 1, r2v1, f2v1, Fs.ctypes.data, Es.ctypes.data, 0, 0, 0,
                                                                                                                   m20.ctypes.data, nm20.ctypes.data), 30), 4),
        "SearchForTriangulation(pKF1, pKF2) cpu_oracle_ms": round(best(lambda: OL.orbo_search_for_triangulation(kf1[0].ref(), kf2[0].ref(), F9.ctypes.data, E2.ctypes.data, 0, 0, 0, mo.ctypes.data), 10), 4),
        "SearchByBoW(pKF, F) gpu_ms": round(best(bow_one, 30), 4),
        "SearchByBoW(pKF, F) gpu_ms_both_resident": round(best(lambda: L.orbm_search_by_bow_resident(ex._h, 1, r1v1, f1v1, r2v1, f2v1, 0.7, 1, 1, po1, nm10.ctypes.data), 30), 4),
        "SearchByBoW(pKF, F) cpu_oracle_ms": round(best(lambda: OL.orbo_search_by_bow(kf1[0].ref(), kf2[0].ref(), 0.7, 1, 1, mo.ctypes.data), 10), 4),
        "note": "a CPU function that takes less than the round-trip floor cannot be beaten by a synchronous device call, whatever the kernel; the batched / neighbour forms amortise the floor"}
    # Tracking::SearchLocalPoints: Frame::isInFrustum for 5000 map points + SearchByProjection on those in view, device vs the reference's own
    # Frame.cc / ORBmatcher.cc (oracle/_ref/libref_frame.so)
    if ol.reference_frame_lib() is not None:
        import test_local_points as tlp
        from orb_slam3_detailed_comments_amd import ComputeStereoMatches, views
        Ls, Rs = synth.stereo_pair(640, 480, seed=7)
        RF = ol.ReferenceFrame(Ls, Rs, 1000, fx=tlp.FX, fy=tlp.FY, cx=tlp.CX, cy=tlp.CY, bf=tlp.BF)
        ex4 = ORBextractor(1000, 1.2, 8, 20, 7)
        (_, kL, dL), _ = ex4.extract_batch(np.stack([Ls, Rs]))
        uu, _, _ = ComputeStereoMatches(ex4, ex4, tlp.BF, RF.mb, 0, 1, 1)
        sfs = ex4.GetScaleFactors()
        fv4 = views.frame_view(kL, dL, sfs, 640, 480, u_right=uu[0, :RF.N], mbf=tlp.BF)
        Rcw = tlp._rot(0.02, -0.03, 0.01); tcw = np.array([0.3, -0.1, 0.25], np.float32)
        sp = tlp._scene(RF, rng, Rcw, tcw, MP)
        g = lambda: M.SearchLocalPoints(ex4, fv4, Rcw, tcw, (tlp.FX, tlp.FY, tlp.CX, tlp.CY), (0.0, 640.0, 0.0, 480.0), tlp.BF, sfs, *sp, 0.5, 3.0, False, 50.0, 0.8)
        c = lambda: RF.search_local_points(Rcw, tcw, *sp, 0.5, True, 3.0, False, 50.0, 0.8)
        same = np.array_equal(g()[1], c()[1])
        graw = M.SearchLocalPoints(ex4, fv4, Rcw, tcw, (tlp.FX, tlp.FY, tlp.CX, tlp.CY), (0.0, 640.0, 0.0, 480.0), tlp.BF, sfs, *sp, 0.5, 3.0, False, 50.0, 0.8, prepared=True)
        rp = M.ResidentPoints(ex4, sp[0], sp[1], sp[2], sp[3], sp[6])
        gres = M.SearchLocalPoints(ex4, fv4, Rcw, tcw, (tlp.FX, tlp.FY, tlp.CX, tlp.CY), (0.0, 640.0, 0.0, 480.0), tlp.BF, sfs, *sp, 0.5, 3.0, False, 50.0, 0.8, prepared=True, resident=rp)
        same_res = np.array_equal(M.SearchLocalPoints(ex4, fv4, Rcw, tcw, (tlp.FX, tlp.FY, tlp.CX, tlp.CY), (0.0, 640.0, 0.0, 480.0), tlp.BF, sfs, *sp, 0.5, 3.0, False, 50.0, 0.8, resident=rp)[1], c()[1])
        out["Tracking::SearchLocalPoints: isInFrustum + SearchByProjection, 5000 map points"] = {"gpu_ms": round(best(g, 5), 3), "gpu_ms_c_abi": round(best(graw, 10), 3),
                                                                                                  "gpu_ms_c_abi_resident_points": round(best(gres, 10), 3),
                                                                                                  "cpu_reference_ms": round(best(c, 2), 3), "identical": bool(same and same_res)}
    # vocabulary transform: k=10, L=5 (111 110 nodes); 128 extracted EuRoC-size images, descriptors resident on the device
    tmp = tempfile.mkdtemp()
    header, parent, leaf, vdesc, weight = vs.make_vocabulary(rng, 10, 5)
    path = os.path.join(tmp, "voc.txt"); vs.write_text(path, header, parent, leaf, vdesc, weight)
    ex2 = ORBextractor(1200, 1.2, 8, 20, 7)
    voc = ORBVocabulary.loadFromTextFile(ex2, path)
    imgs = np.stack([synth.stereo_pair(752, 480, seed=100 + (i % 16))[0] for i in range(128)])
    ex2.enqueue(imgs); res = ex2.fetch()

    def gpu_voc():
        voc.transform_extracted(ex2, 0, 128, 4); ex2.sync()
    t_dev = best(gpu_voc, 5)
    r = voc.fetch(ex2, 0, len(res[0][2]))
    entry = {"gpu_ms_128_images_device_resident": round(t_dev, 3), "gpu_ms_per_image": round(t_dev / 128, 4), "nodes": int(len(parent))}
    if ol.reference_dbow2() is not None:
        ref = ol.RefVocabulary(path)
        d0 = res[0][2]
        entry["cpu_reference_dbow2_ms_per_image"] = round(best(lambda: ref.transform(d0, 4), 3), 3)
        bi, bv, fn, fs, ff = ref.transform(d0, 4)
        entry["parity_image0"] = bool(np.array_equal(r.bow_id, bi) and r.bow_val.tobytes() == bv.tobytes() and np.array_equal(r.fv_feat, ff))
    out["ORBVocabulary::transform (k=10, L=5, 1213 features/image)"] = entry
    # input pre-step: rectification of 128 EuRoC frames (inputs resident in HBM)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_emu_input import rectify_maps
    mx, my = rectify_maps(752, 480, rng)
    ex3 = ORBextractor(1200, 1.2, 8, 20, 7)
    ex3.set_input(1, remap=(mx, my)); ex3.profile(True, serial=True)
    dptr = ex3.device_upload(imgs)
    def gpu_in():
        ex3.enqueue(None, (0, 0), device_ptr=dptr, shape=imgs.shape); ex3.sync()
    best(gpu_in, 3)
    st = ex3.stage_ms()
    out["cv::remap rectification, 128 x 752x480 (k_input_remap -> level 0)"] = {"gpu_ms": round(st["import"], 4), "GBps": round(128 * 752 * 480 * (1 + 8 + 1) / (st["import"] * 1e-3) / 1e9, 1),
                                                                                 "cpu_oracle_ms_per_image": round(best(lambda: ol.oracle_remap(imgs[0], mx, my), 2), 2)}
    # ---- round 4: the remaining per-frame searches in batched device form, BASELINE.json config-4 size: 128 frames of 640x480, 1000 features ----
    import test_local_points as tlp
    Bf = 128
    ex5 = ORBextractor(1000, 1.2, 8, 20, 7)
    imgs5 = np.stack([synth.corner_field(640, 480, seed=400 + (i % 16), nrect=int(3000 * 640 * 480 / (752 * 480))) for i in range(Bf)])
    res5 = ex5.extract_batch(imgs5)
    sfs5 = ex5.GetScaleFactors(); cap5 = ex5.max_keypoints()
    cam5, bounds5 = (tlp.FX, tlp.FY, tlp.CX, tlp.CY), (0.0, 640.0, 0.0, 480.0)
    # relocalisation: SearchByProjection(CurrentFrame, pKF, sAlreadyFound, 10, 100), every frame against the ~1000 map points of its candidate key frame
    capK = cap5
    nK = np.zeros(Bf, np.int32); posK = np.zeros((Bf, capK, 3), np.float32); validK = np.zeros((Bf, capK), np.uint8); mnK = np.zeros((Bf, capK), np.float32)
    mxK = np.zeros((Bf, capK), np.float32); angK = np.zeros((Bf, capK), np.float32); descK = np.zeros((Bf, capK, 32), np.uint8); poses5 = []
    for b in range(Bf):
        k, d = res5[b][1], res5[b][2]; N = len(k); nK[b] = N
        R5, t5 = tlp._rot(*(rng.normal(0, 0.01, 3))), rng.normal(0, 0.05, 3).astype(np.float32); poses5.append((R5, t5))
        z = rng.uniform(1.0, 10.0, N)
        Xc = np.stack([(k["x"] + rng.normal(0, 1.5, N) - tlp.CX) / tlp.FX * z, (k["y"] + rng.normal(0, 1.5, N) - tlp.CY) / tlp.FY * z, z], 1)
        Xw = (R5.astype(np.float64).T @ (Xc - t5.astype(np.float64)).T).T
        posK[b, :N] = Xw; validK[b, :N] = rng.uniform(size=N) < 0.85
        dist = np.linalg.norm(Xw + (R5.astype(np.float64).T @ t5.astype(np.float64)), axis=1)
        mx = dist * 1.2 ** k["octave"].astype(np.float64); mxK[b, :N] = mx; mnK[b, :N] = mx / 1.2 ** 7
        angK[b, :N] = k["angle"]; dd = d.copy()
        flips = rng.integers(0, 256, (N, 12))
        for j in range(12):
            dd[np.arange(N), flips[:, j] >> 3] ^= (1 << (flips[:, j] & 7)).astype(np.uint8)
        descK[b, :N] = dd
    kb = M.KeyFrameBatch(ex5, Bf, cam5, bounds5, tlp.BF, sfs5); kb.set_poses(poses5)

    def reloc():
        kb.enqueue(nK, posK, validK, mnK, mxK, angK, descK, 10.0, 100, True, None); return kb.fetch()
    t_rel = best(reloc, 8); _, nm_rel = reloc()
    out["SearchByProjection(Frame, KeyFrame, sAlreadyFound, 10, 100) x 128 frames, batched on the device (relocalisation)"] = {
        "gpu_ms_per_batch": round(t_rel, 3), "frames_per_s": round(Bf / (t_rel * 1e-3), 1), "avg_matches_per_frame": round(float(nm_rel.mean()), 1), "points_per_key_frame": int(nK.mean())}
    # TrackReferenceKeyFrame / Relocalization: SearchByBoW(pKF, F, vpMapPointMatches), every frame against a resident key frame; vocabulary k = 10, L = 5
    voc5 = ORBVocabulary.loadFromTextFile(ex5, path)
    kfs5, mps5 = [], []
    for b in range(16):                                     # 16 distinct key frames serve the 128 frames (frame b and b + 16 show the same scene)
        k, d = res5[b][1], res5[b][2]
        bw = voc5.transform(descK[b, :len(k)], 4)
        from orb_slam3_detailed_comments_amd import views as V5
        kfs5.append(M.ResidentKeyFrame(ex5, V5.key_frame_view(k, descK[b, :len(k)], sfs5, sfs5 * sfs5, bw.fv_node, bw.fv_start, bw.fv_feat, None, None)))
        mps5.append((rng.uniform(size=len(k)) < 0.85).astype(np.uint8))
    kf_list = [kfs5[b % 16] for b in range(Bf)]; mp_list = [mps5[b % 16] for b in range(Bf)]
    mb5 = M.ORBmatcher(0.7, True)

    def bow_frames():
        voc5.transform_extracted(ex5, 0, Bf, 4)
        return mb5.SearchByBoWFramesBatch(ex5, voc5, kf_list, mp_list)

    def bow_only():
        return mb5.SearchByBoWFramesBatch(ex5, voc5, kf_list, mp_list)
    t_bow = best(bow_frames, 6); t_bow_only = best(bow_only, 6); rb = bow_frames()
    out["SearchByBoW(KeyFrame, Frame) x 128 frames, batched on the device (frames' FeatureVectors from the device vocabulary transform)"] = {
        "gpu_ms_per_batch_with_vocabulary_transform": round(t_bow, 3), "frames_per_s_with_vocabulary_transform": round(Bf / (t_bow * 1e-3), 1),
        "gpu_ms_per_batch_search_only": round(t_bow_only, 3), "frames_per_s_search_only": round(Bf / (t_bow_only * 1e-3), 1),
        "avg_matches_per_frame": round(float(np.mean([r[0] for r in rb])), 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
