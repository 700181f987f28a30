"""Parity tests proper: the hipcc-built library on a real MI355X, through the C ABI, against the CPU oracle
(bit-exact: keypoint records incl. the float angle's bit pattern, descriptors, stereo uRight/depth bit patterns,
kNN indices/distances).  Run with `pytest -m gpu`."""
import os
import threading

import numpy as np
import pytest

import oracle_lib as ol
from cases import FULL_CASES, SMALL_CASES, LARGE_CASES, EUROC_BF, EUROC_B
from orb_slam3_detailed_comments_amd import synth
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
from orb_slam3_detailed_comments_amd import matcher as M

pytestmark = pytest.mark.gpu


def _same(a, b):
    return a[0] == b[0] and ol.kps_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("name,factory,nf,lap", FULL_CASES + SMALL_CASES + LARGE_CASES, ids=[c[0] for c in FULL_CASES + SMALL_CASES + LARGE_CASES])
def test_extractor_vs_oracle(hip_lib, name, factory, nf, lap):
    img = factory()
    ex = ORBextractor(nf, 1.2, 8, 20, 7)
    got = ex(img, None, lap)
    o = ol.OracleExtractor(nf)
    exp = o.extract(img, lap)
    if not _same(got, exp):   # locate the first diverging stage before failing
        for l in range(8):
            assert np.array_equal(ex.pyramid_level(l), o.level_image(l)), "pyramid level %d" % l
            assert np.array_equal(ex.pyramid_level(l, blurred=True), o.level_image(l, blurred=True)), "blur level %d" % l
            assert np.array_equal(ex.debug_candidates(l), o.level_candidates(l)), "FAST candidates level %d" % l
            k2 = o.level_keypoints(l)
            k2a = np.stack([k2["x"] - 16, k2["y"] - 16, k2["response"]], 1).astype(np.int32) if len(k2) else np.zeros((0, 3), np.int32)
            assert np.array_equal(ex.debug_level_keys(l), k2a), "quadtree level %d" % l
    assert _same(got, exp)
    if ol.reference() is not None:      # and against the reference's own source
        assert _same(got, ol.ReferenceExtractor(nf).extract(img, lap))


def test_stagewise_full_size(hip_lib):
    img = synth.corner_field(seed=7)
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    ex(img)
    o = ol.OracleExtractor(1200); o.extract(img)
    for l in range(8):
        assert np.array_equal(ex.pyramid_level(l), o.level_image(l))
        assert np.array_equal(ex.pyramid_level(l, blurred=True), o.level_image(l, blurred=True))
        assert np.array_equal(ex.debug_candidates(l), o.level_candidates(l))


def test_golden_fixtures(hip_lib):
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gd, "gold_376x240_n500.npz"))
    ex = ORBextractor(500, 1.2, 8, 20, 7)
    for key, lap in (("a", (0, 0)), ("b", (100, 250))):
        mono, k, d = ex(g["image"], None, lap)
        assert mono == int(g["mono_" + key]) and k.tobytes() == g["kps_" + key].tobytes() and np.array_equal(d, g["desc_" + key])
    g = np.load(os.path.join(gd, "gold_stereo_376x240_n500.npz"))
    res = ex.extract_batch(np.stack([g["left"], g["right"]]))
    assert res[0][1].tobytes() == g["kps_left"].tobytes() and np.array_equal(res[1][2], g["desc_right"])
    u, d, n = M.ComputeStereoMatches(ex, ex, float(g["bf"]), float(g["b"]), 0, 1, 1)
    N = len(g["uright"])
    assert u[0, :N].tobytes() == g["uright"].tobytes() and d[0, :N].tobytes() == g["depth"].tobytes() and n[0] == int(g["n_matches"])


def test_other_parameters_and_gauss_variant(hip_lib):
    img = synth.corner_field(480, 360, seed=21, nrect=1200)
    for (nf, sf, nl, ini, mn, gv) in [(800, 1.2, 8, 20, 7, 1), (600, 1.5, 4, 25, 10, 0), (700, 1.1, 6, 12, 5, 0), (50, 1.2, 8, 20, 7, 0)]:
        ex = ORBextractor(nf, sf, nl, ini, mn)
        ex.set_gaussian_taps(gv)
        assert _same(ex(img), ol.OracleExtractor(nf, sf, nl, ini, mn, gv).extract(img)), (nf, sf, nl)


def test_empty_and_errors(hip_lib):
    from orb_slam3_detailed_comments_amd._lib import OrbxError
    ex = ORBextractor(500, 1.2, 8, 20, 7)
    assert ex(np.zeros((0, 0), np.uint8))[0] == -1
    with pytest.raises(OrbxError):
        ex(np.zeros((60, 60), np.uint8))
    flat = ex(np.full((480, 752), 90, np.uint8))
    assert flat[0] == 0 and len(flat[1]) == 0          # no corners anywhere: N = 0, monoIndex = 0


def test_stereo_full_size_batch(hip_lib):
    P = 6
    pairs = [synth.stereo_pair(seed=40 + i) for i in range(P)]
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    res = ex.extract_batch(np.stack([p[0] for p in pairs] + [p[1] for p in pairs]))
    u, d, n = M.ComputeStereoMatches(ex, ex, EUROC_BF, EUROC_B, 0, P, P)
    for p in range(P):
        oL, oR = ol.OracleExtractor(1200), ol.OracleExtractor(1200)
        eL, eR = oL.extract(pairs[p][0]), oR.extract(pairs[p][1])
        assert _same(res[p], eL) and _same(res[P + p], eR)
        uo, do, no = ol.oracle_stereo(oL, oR, eL[1], eL[2], eR[1], eR[2], EUROC_BF, EUROC_B)
        N = len(uo)
        assert n[p] == no and no > 100
        assert u[p, :N].tobytes() == uo.tobytes() and d[p, :N].tobytes() == do.tobytes()


def test_stereo_degenerate(hip_lib):
    """No candidates at all (unrelated right image) and identical images (disparity 0 -> 0.01 clamp path)."""
    L = synth.corner_field(seed=60); R = synth.corner_field(seed=61)
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    for (a, b) in ((L, R), (L, L)):
        res = ex.extract_batch(np.stack([a, b]))
        u, d, n = M.ComputeStereoMatches(ex, ex, EUROC_BF, EUROC_B, 0, 1, 1)
        oL, oR = ol.OracleExtractor(1200), ol.OracleExtractor(1200)
        eL, eR = oL.extract(a), oR.extract(b)
        uo, do, no = ol.oracle_stereo(oL, oR, eL[1], eL[2], eR[1], eR[2], EUROC_BF, EUROC_B)
        assert n[0] == no and u[0, :len(uo)].tobytes() == uo.tobytes() and d[0, :len(uo)].tobytes() == do.tobytes()


def test_fisheye_knn_ratio(hip_lib):
    L, R = synth.stereo_pair(512, 512, seed=5)
    ex = ORBextractor(1500, 1.2, 8, 20, 7)
    for lap in ((0, 511), (100, 400)):
        res = ex.extract_batch(np.stack([L, R]), lap)
        out = M.StereoFishEyeKnn(ex, ex, 0, 1, 1)
        (mL, kL, dL), (mR, kR, dR) = res
        assert _same(res[0], ol.OracleExtractor(1500).extract(L, lap))
        ref = ol.oracle_knn2(dL[mL:], dR[mR:])
        nq = len(dL) - mL
        for key in ref:
            assert np.array_equal(out[key][0, :nq], ref[key]), key
        assert ref["ratio_ok"].sum() > 50
    # several pairs per call, the matrix-core kernel (default) against the wave-per-query kernel on the vector units and against the oracle
    pairs = [synth.stereo_pair(512, 512, seed=30 + i) for i in range(3)]
    res = ex.extract_batch(np.stack([p[0] for p in pairs] + [p[1] for p in pairs]), (40, 470))
    out = M.StereoFishEyeKnn(ex, ex, 0, 3, 3)
    ex.debug_stereo_flags(8)
    out2 = M.StereoFishEyeKnn(ex, ex, 0, 3, 3)
    ex.debug_stereo_flags(0)
    for key in out:
        assert np.array_equal(out[key], out2[key]), key
    for i in range(3):
        (mL, kL, dL), (mR, kR, dR) = res[i], res[3 + i]
        ref = ol.oracle_knn2(dL[mL:], dR[mR:])
        for key in ref:
            assert np.array_equal(out[key][i, :len(dL) - mL], ref[key]), (i, key)


def test_hamming_matrix(hip_lib):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (257, 32), dtype=np.uint8); b = rng.integers(0, 256, (1000, 32), dtype=np.uint8)
    a[3] = 0; b[5] = 255; b[7] = a[11]
    ex = ORBextractor(500, 1.2, 8, 20, 7)
    Hm = M.ORBmatcher.DescriptorDistance(ex, a, b)
    assert np.array_equal(Hm, np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2))
    assert Hm[3, 5] == 256 and Hm[11, 7] == 0
    assert all(Hm[i, j] == ol.oracle_hamming(a[i], b[j]) for i in range(0, 257, 31) for j in range(0, 1000, 97))


def test_batch_position_independence_and_idempotence(hip_lib):
    """Size-independent properties at the bench's batch size: an image's result does not depend on its position in the
    batch or on its neighbours; repeating a batch reproduces it bit for bit; device-resident input == host input."""
    B = 32
    imgs = np.stack([synth.corner_field(seed=200 + (i % 5)) for i in range(B)])
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    r1 = ex.extract_batch(imgs)
    for i in range(B):
        assert _same(r1[i], r1[i % 5])
    assert _same(r1[0], ol.OracleExtractor(1200).extract(imgs[0]))
    perm = np.random.default_rng(1).permutation(B)
    r2 = ex.extract_batch(imgs[perm])
    for j, i in enumerate(perm):
        assert _same(r2[j], r1[i])
    dptr = ex.device_upload(imgs)
    ex.enqueue(None, (0, 0), device_ptr=dptr, shape=imgs.shape)
    r3 = ex.fetch()
    assert all(_same(r3[i], r1[i]) for i in range(B))
    ex.device_free(dptr)


def test_two_instances_on_two_threads(hip_lib):
    L, R = synth.stereo_pair(seed=70)
    exL, exR = ORBextractor(1200, 1.2, 8, 20, 7), ORBextractor(1200, 1.2, 8, 20, 7)
    out = {}
    for rep in range(3):
        tl = threading.Thread(target=lambda: out.__setitem__("L", exL(L)))
        tr = threading.Thread(target=lambda: out.__setitem__("R", exR(R)))
        tl.start(); tr.start(); tl.join(); tr.join()
    oL, oR = ol.OracleExtractor(1200), ol.OracleExtractor(1200)
    eL, eR = oL.extract(L), oR.extract(R)
    assert _same(out["L"], eL) and _same(out["R"], eR)
    u, d, n = M.ComputeStereoMatches(exL, exR, EUROC_BF, EUROC_B)
    uo, do, no = ol.oracle_stereo(oL, oR, eL[1], eL[2], eR[1], eR[2], EUROC_BF, EUROC_B)
    assert n[0] == no and u[0, :len(uo)].tobytes() == uo.tobytes()


def test_smoke_entry(hip_lib):
    import __graft_entry__ as g
    g.smoke()


def test_graph_replay_equals_eager(hip_lib):
    """hipGraph replay of the extraction pipeline: same bits as eager launches, across re-use and re-capture."""
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    ex.graph_replay(True)
    for seed, lap in ((80, (0, 0)), (81, (0, 0)), (82, (100, 400)), (83, (0, 0))):
        pair = np.stack(synth.stereo_pair(seed=seed))
        got = ex.extract_batch(pair, lap)
        for i in range(2):
            assert _same(got[i], ol.OracleExtractor(1200).extract(pair[i], lap)), (seed, i)
    u, d, n = M.ComputeStereoMatches(ex, ex, EUROC_BF, EUROC_B, 0, 1, 1)
    assert n[0] > 100
    big = np.stack([synth.corner_field(seed=90 + i) for i in range(5)])        # other batch size -> re-capture
    got = ex.extract_batch(big)
    assert all(_same(got[i], ol.OracleExtractor(1200).extract(big[i])) for i in range(5))


def test_fuzz_gpu(hip_lib):
    from test_emu_fuzz import _case
    for seed in range(14):
        img, nf, sf, nl, ini, mn, lap, gv = _case(seed)
        ex = ORBextractor(nf, sf, nl, ini, mn)
        ex.set_gaussian_taps(gv)
        assert _same(ex(img, None, lap), ol.OracleExtractor(nf, sf, nl, ini, mn, gv).extract(img, lap)), seed


def test_row_ends_of_every_width(hip_lib):
    """Every width residue (mod 4, and the number of columns in the last dword of a row) at every level, raw and blurred: the streaming kernels
    handle row ends by per-thread dword offsets and byte selectors (k_resize_rows, k_blur)."""
    for width in range(321, 337):
        img = synth.uniform_noise(width, 280, seed=300 + width)
        ex = ORBextractor(300, 1.2, 8, 20, 7)
        got = ex(img, None, (0, 0))
        o = ol.OracleExtractor(300)
        exp = o.extract(img, (0, 0))
        for l in range(8):
            assert np.array_equal(ex.pyramid_level(l), o.level_image(l)), "width %d pyramid level %d" % (width, l)
            assert np.array_equal(ex.pyramid_level(l, blurred=True), o.level_image(l, blurred=True)), "width %d blur level %d" % (width, l)
        assert _same(got, exp), width
        ex.close()
