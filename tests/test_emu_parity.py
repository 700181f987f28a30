"""CPU-only CI of the *kernel sources*: orb_slam3_detailed_comments_amd/csrc/k_*.hip compiled with g++ against the
SIMT emulator (tests/emu) must reproduce the oracle bit for bit.  This is not the product path (the package only
loads the hipcc build); the same assertions run against the real GPU in tests/test_gpu_parity.py."""
import threading

import numpy as np
import pytest

import oracle_lib as ol
from cases import SMALL_CASES, FULL_CASES, LARGE_CASES, EUROC_BF, EUROC_B
from orb_slam3_detailed_comments_amd import synth
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
from orb_slam3_detailed_comments_amd import matcher as M


def _same(a, b):
    return a[0] == b[0] and ol.kps_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


# (the full-size and the large cases joined the CPU suite in round 5, when the emulator's fiber switch lost its system calls: 0.2 s per EuRoC image)
ALL_CASES = SMALL_CASES + FULL_CASES + LARGE_CASES


@pytest.mark.parametrize("name,factory,nf,lap", ALL_CASES, ids=[c[0] for c in ALL_CASES])
def test_extractor_stagewise_and_final(emu_lib, name, factory, nf, lap):
    img = factory()
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=emu_lib)
    got = ex(img, None, lap)
    o = ol.OracleExtractor(nf)
    exp = o.extract(img, lap)
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


def test_empty_image_and_errors(emu_lib):
    ex = ORBextractor(500, 1.2, 8, 20, 7, lib=emu_lib)
    mono, k, d = ex(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(k) == 0 and d.shape == (0, 32)        # src/ORBextractor.cc:1561-1562
    from orb_slam3_detailed_comments_amd._lib import OrbxError
    with pytest.raises(OrbxError):                                    # too small for the 35-px cell grid
        ex(np.zeros((60, 60), np.uint8))
    with pytest.raises(OrbxError):
        ORBextractor(0, 1.2, 8, 20, 7, lib=emu_lib)
    with pytest.raises(OrbxError):                                    # orbx_set_pyramid_mode: 0, 1 or 2
        ex.pyramid_mode(3)
    with pytest.raises(OrbxError, match="too small"):                 # one pixel below the smallest size (tests/cases.py: min_239x239)
        ex(np.zeros((238, 238), np.uint8))
    with pytest.raises(OrbxError, match="aspect ratio"):              # taller than twice the width: the reference's nIni is 0 and it divides by it
        ex(np.zeros((479, 239), np.uint8))
    for shape in ((300, 4128), (4128, 300)):                          # one pixel beyond the largest size (tests/cases.py: max_4127x4127 on the GPU)
        with pytest.raises(OrbxError, match="4127"):
            ex(np.zeros(shape, np.uint8))
    assert len(ex(synth.corner_field(376, 240, seed=10, nrect=800))[1]) > 300       # the handle is still usable after the refusals


def test_other_parameters_and_gauss_variant(emu_lib):
    img = synth.corner_field(376, 240, seed=31, nrect=800)
    for (nf, sf, nl, ini, mn, gv) in [(300, 1.2, 8, 20, 7, 1), (250, 1.5, 4, 25, 10, 0), (40, 1.2, 8, 20, 7, 0)]:
        ex = ORBextractor(nf, sf, nl, ini, mn, lib=emu_lib)
        ex.set_gaussian_taps(gv)
        assert _same(ex(img), ol.OracleExtractor(nf, sf, nl, ini, mn, gv).extract(img))


def test_batch_and_stereo_and_knn(emu_lib):
    L, R = synth.stereo_pair(376, 240, seed=20, nrect=800)
    L2, R2 = synth.stereo_pair(376, 240, seed=21, nrect=800)
    ex = ORBextractor(500, 1.2, 8, 20, 7, lib=emu_lib)
    res = ex.extract_batch(np.stack([L, L2, R, R2]))
    u, d, n = M.ComputeStereoMatches(ex, ex, EUROC_BF, EUROC_B, 0, 2, 2)
    for p, (l, r) in enumerate(((L, R), (L2, R2))):
        oL, oR = ol.OracleExtractor(500), ol.OracleExtractor(500)
        eL, eR = oL.extract(l), oR.extract(r)
        assert _same(res[p], eL) and _same(res[2 + p], eR)
        uo, do, no = ol.oracle_stereo(oL, oR, eL[1], eL[2], eR[1], eR[2], EUROC_BF, EUROC_B)
        N = len(eL[1])
        assert n[p] == no and no > 50
        assert u[p, :N].tobytes() == uo.tobytes() and d[p, :N].tobytes() == do.tobytes()
        assert (u[p, N:] == -1).all()
    # fisheye-style kNN + ratio on the lapping part
    res = ex.extract_batch(np.stack([L, R]), (60, 300))
    out = M.StereoFishEyeKnn(ex, ex, 0, 1, 1)
    (mL, kL, dL), (mR, kR, dR) = res
    ref = ol.oracle_knn2(dL[mL:], dR[mR:])
    nq = len(dL) - mL
    assert nq > 100 and mL > 0
    for key in ref:
        assert np.array_equal(out[key][0, :nq], ref[key]), key
    # the same through the wave-per-query kernel on the vector units (the default is the distance matrix on the matrix cores, k_knn2_mfma)
    ex.debug_stereo_flags(8)
    out2 = M.StereoFishEyeKnn(ex, ex, 0, 1, 1)
    ex.debug_stereo_flags(0)
    for key in out:
        assert np.array_equal(out[key], out2[key]), key
    # all-pairs DescriptorDistance
    Hm = M.ORBmatcher.DescriptorDistance(ex, dL[:40], dR[:50])
    assert np.array_equal(Hm, np.unpackbits(dL[:40, None, :] ^ dR[None, :50, :], axis=2).sum(2))


def test_knn_ties_and_degenerate_sizes(emu_lib):
    ex = ORBextractor(500, 1.2, 8, 20, 7, lib=emu_lib)
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (6, 32), dtype=np.uint8)
    dup = np.concatenate([a, a, a[:2]])             # duplicate descriptors: equal distances keep the lower index
    Hm = M.ORBmatcher.DescriptorDistance(ex, dup, dup)
    assert (np.diag(Hm) == 0).all() and Hm[0, 6] == 0
    ref = ol.oracle_knn2(dup, dup)
    assert ref["idx0"][6] == 0 and ref["idx1"][6] == 6 and ref["dist1"][6] == 0      # strict '<' insertion order


def test_two_extractor_instances_concurrently(emu_lib):
    """Two instances on two threads, like Frame's left/right extraction threads (src/Frame.cc:136-141)."""
    L, R = synth.stereo_pair(376, 240, seed=22, nrect=800)
    exL, exR = ORBextractor(500, 1.2, 8, 20, 7, lib=emu_lib), ORBextractor(500, 1.2, 8, 20, 7, lib=emu_lib)
    out = {}
    tl = threading.Thread(target=lambda: out.__setitem__("L", exL(L)))
    tr = threading.Thread(target=lambda: out.__setitem__("R", exR(R)))
    tl.start(); tr.start(); tl.join(); tr.join()
    assert _same(out["L"], ol.OracleExtractor(500).extract(L)) and _same(out["R"], ol.OracleExtractor(500).extract(R))
    u, d, n = M.ComputeStereoMatches(exL, exR, EUROC_BF, EUROC_B)
    oL, oR = ol.OracleExtractor(500), ol.OracleExtractor(500)
    eL, eR = oL.extract(L), oR.extract(R)
    uo, do, no = ol.oracle_stereo(oL, oR, eL[1], eL[2], eR[1], eR[2], EUROC_BF, EUROC_B)
    assert n[0] == no and u[0, :len(uo)].tobytes() == uo.tobytes()


def test_golden_fixture(emu_lib):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gold_376x240_n500.npz"))
    ex = ORBextractor(500, 1.2, 8, 20, 7, lib=emu_lib)
    mono, k, d = ex(g["image"], None, (100, 250))
    assert mono == int(g["mono_b"]) and k.tobytes() == g["kps_b"].tobytes() and np.array_equal(d, g["desc_b"])


def test_handle_reuse_across_sizes_and_batches(emu_lib):
    """One handle, changing resolution and batch size between calls: tables and device buffers are rebuilt transparently."""
    ex = ORBextractor(300, 1.2, 8, 20, 7, lib=emu_lib)
    a = synth.corner_field(376, 240, seed=40, nrect=800); b = synth.corner_field(320, 300, seed=41, nrect=800)
    ra1 = ex(a)
    rb = ex.extract_batch(np.stack([b, b[::-1].copy(), b]))
    ra2 = ex(a)
    assert _same(ra1, ra2) and _same(ra1, ol.OracleExtractor(300).extract(a))
    assert _same(rb[0], rb[2]) and _same(rb[0], ol.OracleExtractor(300).extract(b)) and _same(rb[1], ol.OracleExtractor(300).extract(b[::-1].copy()))
    # strided input (a view into a wider buffer), like a cv::Mat ROI
    wide = np.zeros((240, 500), np.uint8); wide[:, 60:436] = a
    roi = wide[:, 60:436]
    mono = np.zeros(1, np.int32)
    import ctypes as C
    cap = ex.max_keypoints()
    from orb_slam3_detailed_comments_amd._lib import KP_DTYPE
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = C.c_int(); m = C.c_int()
    emu_lib.check(emu_lib.L.orbx_extract(ex._h, roi.ctypes.data, 376, 240, 500, 0, 0, kps.ctypes.data, desc.ctypes.data, cap, C.byref(n), C.byref(m)))
    assert m.value == ra1[0] and ol.kps_equal(kps[:n.value], ra1[1]) and np.array_equal(desc[:n.value], ra1[2])


def test_unusual_pyramids_with_stereo(emu_lib):
    """Pyramids the reference accepts but nobody ships: 11 levels (stereo row bands up to 25 rows tall), 3 levels at scale 2, one level."""
    from orb_slam3_detailed_comments_amd import matcher as M
    L, R = synth.stereo_pair(752, 480, seed=3)
    for (nf, sf, nl) in [(800, 1.2, 11), (300, 2.0, 3), (2000, 1.3, 1)]:
        ex = ORBextractor(nf, sf, nl, 20, 7, lib=emu_lib)
        res = ex.extract_batch(np.stack([L, R]), (0, 0))
        oL = ol.OracleExtractor(nf, sf, nl, 20, 7); oR = ol.OracleExtractor(nf, sf, nl, 20, 7)
        eL = oL.extract(L); eR = oR.extract(R)
        for got, exp in zip(res, (eL, eR)):
            assert got[0] == exp[0] and ol.kps_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]), (nf, sf, nl)
        u, d, n = M.ComputeStereoMatches(ex, ex, EUROC_BF, EUROC_B, 0, 1, 1)
        uo, do, no = ol.oracle_stereo(oL, oR, eL[1], eL[2], eR[1], eR[2], EUROC_BF, EUROC_B)
        N = len(eL[1])
        assert n[0] == no and np.array_equal(u[0, :N].view(np.uint32), uo.view(np.uint32)) and np.array_equal(d[0, :N].view(np.uint32), do.view(np.uint32)), (nf, sf, nl)
        assert no > 20


@pytest.mark.parametrize("ini,mn", [(0, 0), (1, 1), (20, 0), (7, 20), (2, 1), (130, 128), (254, 1), (255, 255), (90, 89)])
def test_extreme_fast_thresholds(emu_lib, ini, mn):
    """thresholds at the ends of the byte range, minTh above iniTh, threshold 0 (every pixel takes both polarities in k_fast_cells) - against the
    oracle and the reference build"""
    for img in (synth.uniform_noise(200, 180, seed=77), synth.corner_field(260, 200, seed=78, nrect=300)):
        got = ORBextractor(200, 1.2, 4, ini, mn, lib=emu_lib)(img)
        assert _same(got, ol.OracleExtractor(200, 1.2, 4, ini, mn, 0).extract(img)), (ini, mn)
        if ol.reference() is not None:
            assert _same(got, ol.ReferenceExtractor(200, 1.2, 4, ini, mn, 0).extract(img)), (ini, mn)


def _zero_copy_case(lib, w, h, nf):
    """frames written straight into pyramid level 0 (orbx_input_buffer / orbx_input_upload) and extracted in place give what the import path gives"""
    imgs = np.stack([synth.corner_field(w, h, seed=90 + s, nrect=max(200, w * h // 200)) for s in range(3)])
    a = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib); b = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    ref = a.extract_batch(imgs)
    ptr, shape, stride, istride = b.input_upload(imgs)
    assert stride >= w and stride % 64 == 0
    for _ in range(2):                                   # the frames stay resident: a second extraction reads the same level 0
        b.enqueue(None, (0, 0), device_ptr=ptr, shape=shape, stride=stride, image_stride=istride)
        got = b.fetch()
        for x, y in zip(got, ref):
            assert _same(x, y)
    assert np.array_equal(b.pyramid_level(0, 1), imgs[1]) and np.array_equal(b.pyramid_level(3, 2), a.pyramid_level(3, 2))
    a.close(); b.close()


def test_zero_copy_input_emulated(emu_lib):
    _zero_copy_case(emu_lib, 376, 240, 400)
    _zero_copy_case(emu_lib, 320, 256, 300)               # width = pitch (no row padding)


@pytest.mark.gpu
def test_zero_copy_input_gpu(hip_lib):
    _zero_copy_case(hip_lib, 752, 480, 1200)
    _zero_copy_case(hip_lib, 512, 512, 1500)


def _garbage_padding_case(lib, w, h, nf):
    """A producer that copies full-pitch rows leaves arbitrary bytes in the row padding [w, pitch) of level 0; the kernels load them as parts of
    dwords (k_pyramid_fused, k_resize_rows, the FAST window loads) but no output may depend on them (include/orbx.h: orbx_input_buffer)."""
    imgs = np.stack([synth.corner_field(w, h, seed=95 + s, nrect=max(200, w * h // 200)) for s in range(2)])
    a = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib); b = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    ref = a.extract_batch(imgs)
    ptr, shape, stride, istride = b.input_upload(imgs)
    assert stride > w, "this case needs row padding"
    raw = b.pinned_empty((2, istride), np.uint8)
    raw[:] = np.random.default_rng(3).integers(0, 256, raw.shape, dtype=np.uint8)           # garbage everywhere, rows included ...
    for i in range(2):
        rows = raw[i, :h * stride].reshape(h, stride)
        rows[:, :w] = imgs[i]                                                                # ... then the pixels where they belong
    b.device_upload_async(ptr, raw)
    b.enqueue(None, (0, 0), device_ptr=ptr, shape=shape, stride=stride, image_stride=istride)
    got = b.fetch()
    for x, y in zip(got, ref):
        assert _same(x, y)
    for l in range(8):
        assert np.array_equal(b.pyramid_level(l, 1), a.pyramid_level(l, 1)), "level %d" % l
    a.close(); b.close()


def _async_upload_pipeline_case(lib, w, h, nf, rounds):
    """A producer that uploads the next pair into level 0 (orbx_device_upload_async, copy stream) right behind the enqueue of the current one: the
    upload has to wait, on the device, until the extraction in flight has read its frames - for small batches the event it waits for is recorded
    on demand, behind the whole chain (orbx_internal.h).  Three different pairs in turn; a premature overwrite shows up as another pair's keypoints."""
    sets = [np.stack(synth.stereo_pair(w, h, seed=700 + k, nrect=max(300, w * h // 200))) for k in range(3)]
    ref_ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    refs = [ref_ex.extract_batch(x) for x in sets]
    ref_ex.close()
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    ptr, shape, stride, istride = ex.input_upload(sets[0])
    stage = []
    for k in range(3):
        raw = ex.pinned_empty((2, istride), np.uint8); raw[:] = 0
        for i in range(2):
            raw[i, :h * stride].reshape(h, stride)[:, :w] = sets[k][i]
        stage.append(raw)
    ex.device_upload_async(ptr, stage[0])
    for it in range(rounds):
        ex.enqueue(None, (0, 0), device_ptr=ptr, shape=shape, stride=stride, image_stride=istride)
        ex.device_upload_async(ptr, stage[(it + 1) % 3])          # the next pair, while this one is being extracted
        got = ex.fetch()
        for x, y in zip(got, refs[it % 3]):
            assert _same(x, y), "round %d" % it
    ex.close()


def test_async_upload_pipeline_emulated(emu_lib):
    _async_upload_pipeline_case(emu_lib, 376, 240, 400, 4)


@pytest.mark.gpu
def test_async_upload_pipeline_gpu(hip_lib):
    _async_upload_pipeline_case(hip_lib, 752, 480, 1200, 60)


def test_zero_copy_input_garbage_padding_emulated(emu_lib):
    _garbage_padding_case(emu_lib, 376, 240, 400)
    _garbage_padding_case(emu_lib, 330, 260, 300)


@pytest.mark.gpu
def test_zero_copy_input_garbage_padding_gpu(hip_lib):
    _garbage_padding_case(hip_lib, 752, 480, 1200)


@pytest.mark.parametrize("width", list(range(321, 337)))
def test_row_ends_of_every_width(emu_lib, width):
    """The streaming kernels (k_resize_rows, k_blur) handle the end of a row by per-thread dword offsets and byte selectors that depend on
    width mod 4 and on how many columns of the last dword exist: every residue at every level, raw and blurred, against the oracle."""
    img = synth.uniform_noise(width, 280, seed=300 + width)
    ex = ORBextractor(300, 1.2, 8, 20, 7, lib=emu_lib)
    got = ex(img, None, (0, 0))
    o = ol.OracleExtractor(300)
    exp = o.extract(img, (0, 0))
    for l in range(8):
        assert np.array_equal(ex.pyramid_level(l), o.level_image(l)), "pyramid level %d" % l
        assert np.array_equal(ex.pyramid_level(l, blurred=True), o.level_image(l, blurred=True)), "blur level %d" % l
    assert _same(got, exp)


def _pyramid_modes_case(lib, shapes):
    """ComputePyramid as one launch (k_pyramid_fused: tiles of the top level, each level partitioned among them) against one launch per level
    and against the oracle: every level, every byte"""
    for (w, h, sf, nl, nimg) in shapes:
        imgs = np.stack([synth.pink_noise(w, h, seed=300 + 7 * s + w, beta=1.0) for s in range(nimg)])
        levels = {}
        for mode in (1, 2):
            ex = ORBextractor(200, sf, nl, 20, 7, lib=lib)
            ex.pyramid_mode(mode)
            res = ex.extract_batch(imgs)
            levels[mode] = [[ex.pyramid_level(l, b) for l in range(nl)] for b in range(nimg)]
            levels[(mode, "res")] = res
            ex.close()
        for b in range(nimg):
            o = ol.OracleExtractor(200, sf, nl, 20, 7)
            exp = o.extract(imgs[b])
            for l in range(nl):
                assert np.array_equal(levels[2][b][l], o.level_image(l)), ("fused vs oracle", w, h, sf, nl, b, l)
                assert np.array_equal(levels[1][b][l], levels[2][b][l]), ("fused vs per level", w, h, sf, nl, b, l)
            assert _same(levels[(2, "res")][b], exp) and _same(levels[(1, "res")][b], exp)


def test_pyramid_one_launch_emulated(emu_lib):
    # (width, height, scale factor, levels, images): widths that are no multiple of 4 on any level, a factor of 2 (taps two apart), more
    # levels than the reference uses, a single level, two levels, factor 1.5
    _pyramid_modes_case(emu_lib, [(301, 277, 1.2, 8, 2), (422, 360, 2.0, 3, 1), (600, 520, 1.2, 11, 1), (210, 190, 1.2, 1, 1), (263, 251, 1.5, 2, 3), (417, 233, 1.5, 4, 1), (431, 397, 2.5, 2, 2), (530, 470, 3.0, 2, 1),   # factors > 2: source pixels no tap reads are still owned and written
                                   (900, 860, 1.2, 14, 1)])   # 14 levels: a tile's level-0 window no longer fits the LDS budget, the library keeps one launch per level


@pytest.mark.gpu
def test_pyramid_one_launch_gpu(hip_lib):
    _pyramid_modes_case(hip_lib, [(752, 480, 1.2, 8, 2), (301, 277, 1.2, 8, 3), (644, 400, 2.0, 3, 1), (1241, 376, 1.2, 8, 1), (1001, 841, 1.2, 11, 1), (263, 251, 1.5, 2, 5), (1920, 1080, 1.2, 8, 1)])


def _blur_strip_rows_case(lib, shapes):
    """batches above 32 images blur strips of 32 rows (k_blur_large), smaller ones strips of 16 (k_blur / the fused launch): the blurred pyramids of the same
    image must be byte-equal in both and equal to the oracle's - heights that leave partial strips and partial row pairs, widths with partial dwords,
    both OpenCV tap generations"""
    for (w, h, gv) in shapes:
        img = synth.pink_noise(w, h, seed=w + h)
        other = synth.sparse_corners(w, h, seed=5, ncorner=30)
        o = ol.OracleExtractor(300, 1.2, 8, 20, 7, gv)
        o.extract(img)
        ex = ORBextractor(300, 1.2, 8, 20, 7, lib=lib)
        ex.set_gaussian_taps(gv)
        ex.extract_batch(np.stack([other] * 32 + [img]))             # 33 images: 32-row strips; the image under test is the last one
        large = [ex.pyramid_level(l, 32, blurred=True) for l in range(8)]
        ex(img)
        for l in range(8):
            exp = o.level_image(l, blurred=True)
            assert np.array_equal(large[l], exp), "32-row strips, level %d of %dx%d" % (l, w, h)
            assert np.array_equal(ex.pyramid_level(l, blurred=True), exp), "16-row strips, level %d of %dx%d" % (l, w, h)


def test_blur_strip_rows_emulated(emu_lib):
    _blur_strip_rows_case(emu_lib, [(301, 277, 0), (376, 240, 1), (333, 257, 0)])


@pytest.mark.gpu
def test_blur_strip_rows_gpu(hip_lib):
    _blur_strip_rows_case(hip_lib, [(752, 480, 0), (301, 277, 1), (1241, 376, 0), (515, 513, 0)])


def _launch_forms_case(lib, shapes):
    """the launch forms of small batches (blur strips + FAST cells in one launch on one stream, no event records inside the chain:
    orbx_set_small_batch_forms) against the large-batch forms of the same kernels on the same images, and against the oracle: blurred pyramid,
    keypoints, descriptors, stereo matches; then a second handle as the right camera (the left one waits for the right one's ev_done, which is
    recorded on demand)"""
    for (w, h, nf, pairs) in shapes:
        Ls, Rs = zip(*[synth.stereo_pair(w, h, seed=500 + 3 * s + w, nrect=max(300, w * h // 200)) for s in range(pairs)])
        batch = np.stack(list(Ls) + list(Rs))
        out = {}
        for on in (True, False):
            ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
            ex.set_small_batch_forms(on)
            res = ex.extract_batch(batch)
            u, d, n = M.ComputeStereoMatches(ex, ex, EUROC_BF, EUROC_B, 0, pairs, pairs)
            out[on] = (res, u.copy(), d.copy(), n.copy(), [[ex.pyramid_level(l, b, blurred=True) for l in range(8)] for b in range(2 * pairs)])
            ex.close()
        for b in range(2 * pairs):
            assert _same(out[True][0][b], out[False][0][b]), (w, h, b)
            o = ol.OracleExtractor(nf)
            assert _same(out[True][0][b], o.extract(batch[b])), (w, h, b)
            for l in range(8):
                assert np.array_equal(out[True][4][b][l], out[False][4][b][l]) and np.array_equal(out[True][4][b][l], o.level_image(l, blurred=True)), (w, h, b, l)
        for k in (1, 2, 3):
            assert out[True][k].tobytes() == out[False][k].tobytes(), (w, h, k)
        assert out[True][3].min() > 20
        # two handles: left and right extracted by different extractors, the stereo search of the left one waits for the right one
        exL, exR = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib), ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
        exL.extract_batch(np.stack(Ls)); exR.extract_batch(np.stack(Rs))
        u2, d2, n2 = M.ComputeStereoMatches(exL, exR, EUROC_BF, EUROC_B, 0, 0, pairs)
        assert u2.tobytes() == out[True][1].tobytes() and d2.tobytes() == out[True][2].tobytes() and n2.tobytes() == out[True][3].tobytes()
        exL.close(); exR.close()


def test_small_batch_launch_forms_emulated(emu_lib):
    _launch_forms_case(emu_lib, [(376, 240, 500, 1), (301, 277, 300, 2)])


@pytest.mark.gpu
def test_small_batch_launch_forms_gpu(hip_lib):
    _launch_forms_case(hip_lib, [(752, 480, 1200, 1), (376, 240, 500, 3), (1241, 376, 2000, 1), (640, 480, 1000, 16)])


def _zero_rows_case(lib, w, h, nf):
    """descriptor rows beyond an image's keypoint count read as zero in the device-resident block, also when the handle's previous batch filled
    them (no fill launch in front of an extraction: k_orient_brief clears one row per unused keypoint slot).  The fetch with the device's own
    layout copies the whole block, so the page-locked staging array shows every row."""
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    dense = np.stack([synth.corner_field(w, h, seed=400 + s, nrect=max(300, w * h // 150)) for s in range(3)])
    sparse = np.stack([synth.sparse_corners(w, h, seed=410, ncorner=9), np.full((h, w), 90, np.uint8), synth.threshold_blocks(w, h, seed=411)])
    for batch in (dense, sparse, dense[:2], sparse[1:]):
        res = ex.extract_batch(batch)
        block = ex._fetch_buf[2]
        assert block.shape == (len(batch), ex.max_keypoints(), 32)
        for b, (_, k, desc) in enumerate(res):
            assert np.array_equal(block[b, :len(k)], desc) and not block[b, len(k):].any(), (b, len(k))
    assert len(res[-1][1]) > 0 and len(ex.extract_batch(dense)[0][1]) > 100
    ex.close()


def test_descriptor_rows_beyond_count_are_zero_emulated(emu_lib):
    _zero_rows_case(emu_lib, 376, 240, 500)


@pytest.mark.gpu
def test_descriptor_rows_beyond_count_are_zero_gpu(hip_lib):
    _zero_rows_case(hip_lib, 752, 480, 1200)


def test_feature_counts_beyond_the_lds_take_the_node_pool(emu_lib, monkeypatch):
    """The quadtree of a level lives in LDS (81 bytes per node) when it fits: with gfx950's 160 KB a single 752x480 image takes nfeatures = 7800 in LDS on
    every level; 8000 (refused until round 6) runs its level 0 in the global node pool, a batch of 40 (the narrow-counter form) still in LDS; the
    emulator is given the device's LDS size for this test.  All of them are the reference's result (tests/test_mono_init.py: the shipped settings)."""
    monkeypatch.setenv("ORBX_EMU_LDS_LIMIT", "163840")
    img = synth.uniform_noise(752, 480, seed=3)
    ex = ORBextractor(8000, 1.2, 8, 20, 7, lib=emu_lib)
    exp = ol.OracleExtractor(8000).extract(img)
    got = ex(img)
    assert ex.debug_quadtree_pool_levels() == 1
    assert ol.kps_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]) and len(got[1]) > 7000
    sparse = synth.sparse_corners(752, 480, seed=4, ncorner=60)          # (34 noise images would take the emulator a minute)
    got = ex.extract_batch(np.stack([sparse] * 33 + [img]))             # > ORBX_QT_WIDE_BATCH images: 16-byte counters
    assert ex.debug_quadtree_pool_levels() == 0
    for g, im in ((got[0], sparse), (got[33], img)):
        e2 = ol.OracleExtractor(8000).extract(im)
        assert ol.kps_equal(g[1], e2[1]) and np.array_equal(g[2], e2[2])
    ex2 = ORBextractor(7800, 1.2, 8, 20, 7, lib=emu_lib)
    ok = ex2(img)
    exp = ol.OracleExtractor(7800).extract(img)
    assert ex2.debug_quadtree_pool_levels() == 0 and ol.kps_equal(ok[1], exp[1]) and np.array_equal(ok[2], exp[2])

