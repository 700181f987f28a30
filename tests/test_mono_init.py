"""Every extractor Tracking constructs runs through the drop-in, bit for bit - including mpIniORBextractor = ORBextractor(5 * nFeatures, ..) of the
monocular sensors (src/Tracking.cc:634-635, :1331-1332), whose first quadtree levels outgrow the 160 KB of LDS and keep their node lists in the global
node pool (k_quadtree_spill; the reference's std::list has no bound, src/ORBextractor.cc:711-1057).

tests/golden/mono_settings.json holds the extractor parameters of all monocular / monocular-inertial settings files the reference ships
(tools/gen_mono_settings.py).  Checked: the library against the reference's own ORBextractor.cc (oracle/_ref/libref_orb.so) on corner-field AND natural
images for every distinct (size, 5 x nFeatures); the C++ facade (include/orb_slam3_amd/ORBextractor.h) against the reference build of the same driver;
the pool form forced onto the ordinary cases (it must not change a bit); batches on either side of the counter-width switch."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from cases import SMALL_CASES, FULL_CASES
from orb_slam3_detailed_comments_amd import synth, _lib
from orb_slam3_detailed_comments_amd.extractor import ORBextractor

ROOT = ol.ROOT
SETTINGS = json.load(open(os.path.join(ROOT, "tests", "golden", "mono_settings.json")))
# distinct (width, height, 5 * nFeatures, scaleFactor, nLevels, iniThFAST, minThFAST) with the files that ask for it
INIT = {}
for _r in SETTINGS:
    INIT.setdefault((_r["width"], _r["height"], 5 * _r["nFeatures"], _r["scaleFactor"], _r["nLevels"], _r["iniThFAST"], _r["minThFAST"]), []).append(_r["file"])
INIT_KEYS = sorted(INIT)
INIT_IDS = ["%dx%d_n%d" % k[:3] for k in INIT_KEYS]
GFX950_LDS = "163840"
REF_DRIVER = os.path.join(ROOT, "oracle", "_ref", "facade_driver_ref")


def _same(a, b):
    return a[0] == b[0] and ol.kps_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def _expected(nf, sf, nl, ini, mn, img, lap):
    if ol.reference() is not None:
        return ol.ReferenceExtractor(nf, sf, nl, ini, mn).extract(img, lap)
    return ol.OracleExtractor(nf, sf, nl, ini, mn).extract(img, lap)


def test_settings_fixture_covers_the_shipped_files():
    assert len(SETTINGS) >= 50 and (1241, 376, 10000, 1.2, 8, 20, 7) in INIT and (512, 512, 7500, 1.2, 8, 20, 7) in INIT
    if os.path.isdir("/root/reference/Examples"):          # in the build container: the fixture is what the generator writes today
        import glob
        files = {os.path.relpath(f, "/root/reference") for f in glob.glob("/root/reference/Examples*/Monocular*/**/*.yaml", recursive=True)}
        assert {r["file"] for r in SETTINGS} == files


def _init_case(lib, key):
    w, h, nf, sf, nl, ini, mn = key
    ex = ORBextractor(nf, sf, nl, ini, mn, lib=lib)
    # corner field: every level fills its quota (the tree runs to its final rounds); natural: few corners on the top levels
    for img, lap in ((synth.corner_field(w, h, seed=40, nrect=int(3000 * w * h / (752 * 480))), (0, 1000)), (synth.natural(w, h, seed=41), (0, 0))):
        got = ex(img, None, lap)
        assert _same(got, _expected(nf, sf, nl, ini, mn, img, lap)), "%dx%d nfeatures %d (%s)" % (w, h, nf, INIT[key][0])
    return ex


@pytest.mark.parametrize("key", INIT_KEYS, ids=INIT_IDS)
def test_mono_init_extractor_emulated(emu_lib, monkeypatch, key):
    monkeypatch.setenv("ORBX_EMU_LDS_LIMIT", GFX950_LDS)       # the emulator decides LDS form / pool form per level like the device does
    _init_case(emu_lib, key)


@pytest.mark.gpu
@pytest.mark.parametrize("key", INIT_KEYS, ids=INIT_IDS)
def test_mono_init_extractor_gpu(hip_lib, key):
    _init_case(hip_lib, key)


# ... and mpORBextractorLeft = ORBextractor(nFeatures, ..) of the same files (what runs after initialisation): every distinct (size, nFeatures), same two image kinds
POST = sorted({(k[0], k[1], k[2] // 5) + k[3:] for k in INIT_KEYS})
POST_IDS = ["%dx%d_n%d" % k[:3] for k in POST]


@pytest.mark.parametrize("key", POST, ids=POST_IDS)
def test_mono_extractor_after_initialisation_emulated(emu_lib, monkeypatch, key):
    monkeypatch.setenv("ORBX_EMU_LDS_LIMIT", GFX950_LDS)
    INIT.setdefault(key, ["(nFeatures of " + INIT[(key[0], key[1], 5 * key[2]) + key[3:]][0] + ")"])
    assert _init_case(emu_lib, key).debug_quadtree_pool_levels() == 0          # every level of the post-initialisation extractors fits the LDS


@pytest.mark.gpu
@pytest.mark.parametrize("key", POST, ids=POST_IDS)
def test_mono_extractor_after_initialisation_gpu(hip_lib, key):
    INIT.setdefault(key, ["(nFeatures of " + INIT[(key[0], key[1], 5 * key[2]) + key[3:]][0] + ")"])
    assert _init_case(hip_lib, key).debug_quadtree_pool_levels() == 0


def test_pool_form_is_what_the_big_settings_take(emu_lib, monkeypatch):
    """KITTI's 10 000 and TUM-VI's 7 500 need more LDS than gfx950 has in the LDS form - the refusal of the earlier revisions - so they do exercise the pool
    form above, and (2000, KITTI) after initialisation does not."""
    monkeypatch.setenv("ORBX_EMU_LDS_LIMIT", GFX950_LDS)
    for (w, h, nf, pool_levels) in ((1241, 376, 10000, True), (512, 512, 7500, True), (1241, 376, 2000, False), (752, 480, 1200, False)):
        ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=emu_lib)
        ex(synth.sparse_corners(w, h, seed=4, ncorner=60))
        assert (ex.debug_quadtree_pool_levels() > 0) == pool_levels, (w, h, nf)


FORCED = SMALL_CASES + FULL_CASES


def _forced_case(lib, factory, nf, lap, limit):
    img = factory()
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    ex.debug_quadtree_lds_nodes(limit)
    got = ex(img, None, lap)
    assert ex.debug_quadtree_pool_levels() == (8 if limit == 0 else ex.debug_quadtree_pool_levels()) and ex.debug_quadtree_pool_levels() > 0
    assert _same(got, _expected(nf, 1.2, 8, 20, 7, img, lap))
    ex.debug_quadtree_lds_nodes(4000)                          # and back: the handle re-plans
    assert _same(ex(img, None, lap), got) and ex.debug_quadtree_pool_levels() == 0


@pytest.mark.parametrize("name,factory,nf,lap", FORCED, ids=[c[0] for c in FORCED])
def test_pool_form_forced_emulated(emu_lib, name, factory, nf, lap):
    _forced_case(emu_lib, factory, nf, lap, 0)


def test_pool_form_mixed_launch_emulated(emu_lib):
    """levels 0-2 in the pool, the rest in LDS (two launches), 1 image and 34 images (the narrow counters of the large-batch form)"""
    img = synth.corner_field(376, 240, seed=10, nrect=800)
    other = synth.uniform_noise(376, 240, seed=3)
    ex = ORBextractor(500, 1.2, 8, 20, 7, lib=emu_lib)
    ex.debug_quadtree_lds_nodes(85)                            # quotas 109, 91, 75, ...: kp_cap + 8 = 120, 102, 86 | 74 ...
    exp, expo = _expected(500, 1.2, 8, 20, 7, img, (0, 0)), _expected(500, 1.2, 8, 20, 7, other, (0, 0))
    assert _same(ex(img), exp) and ex.debug_quadtree_pool_levels() == 3
    got = ex.extract_batch(np.stack([img] * 17 + [other] * 17))
    assert _same(got[0], exp) and _same(got[16], exp) and _same(got[17], expo) and _same(got[33], expo)


@pytest.mark.gpu
@pytest.mark.parametrize("name,factory,nf,lap", FORCED, ids=[c[0] for c in FORCED])
def test_pool_form_forced_gpu(hip_lib, name, factory, nf, lap):
    _forced_case(hip_lib, factory, nf, lap, 0)


@pytest.mark.gpu
def test_pool_form_batches_gpu(hip_lib):
    """the monocular initialisation extractor on batches: 1, 8 (wide trees, 32-byte counters) and 40 images (four waves per tree, 16-byte counters)"""
    imgs = [synth.corner_field(1241, 376, seed=50 + i, nrect=3900) for i in range(3)] + [synth.natural(1241, 376, seed=60)]
    exp = [_expected(10000, 1.2, 8, 20, 7, im, (0, 0)) for im in imgs]
    ex = ORBextractor(10000, 1.2, 8, 20, 7, lib=hip_lib)
    for B in (1, 8, 40):
        got = ex.extract_batch(np.stack([imgs[i % 4] for i in range(B)]))
        assert ex.debug_quadtree_pool_levels() > 0
        for i in range(B):
            assert _same(got[i], exp[i % 4]), (B, i)
    ex.graph_replay(True)                                      # the two quadtree launches (pool + LDS) inside a captured hipGraph, replayed
    for _ in range(3):
        got = ex.extract_batch(np.stack(imgs[:2]))
        assert _same(got[0], exp[0]) and _same(got[1], exp[1]) and ex.debug_quadtree_pool_levels() > 0


# ---- the C++ facade: ORBextractor(5 * nFeatures, ...) for every settings file, against the reference build of the same driver ----
def _facade_exe(tmp, libdir, libname):
    exe = tmp / "facade_driver_ours"
    subprocess.run(["g++", "-std=c++14", "-O1", "-w", "-I" + os.path.join(ROOT, "include", "orb_slam3_amd"), "-I" + os.path.join(ROOT, "oracle", "opencv_shim"),
                    os.path.join(ROOT, "tests", "cpp", "facade_driver.cpp"), "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    return exe


def _facade_all(tmp, exe, env):
    sizes = sorted({k[:3] for k in INIT_KEYS if k[3:] == (1.2, 8, 20, 7)})           # the driver constructs (n, 1.2, 8, 20, 7): every shipped file asks for that
    assert len(sizes) == len(INIT_KEYS)
    for (w, h, nf) in sizes:
        img = synth.corner_field(w, h, seed=44, nrect=int(3000 * w * h / (752 * 480)))
        raw = tmp / "im.raw"; raw.write_bytes(img.tobytes())
        a, b = tmp / "ref.bin", tmp / "ours.bin"
        args = [str(raw), str(w), str(h), str(nf), "0", "0"]
        subprocess.run([REF_DRIVER] + args + [str(a)], check=True)
        subprocess.run([str(exe)] + args + [str(b)], check=True, env=env)
        assert a.read_bytes() == b.read_bytes(), "facade ORBextractor(%d) at %dx%d differs from the reference build" % (nf, w, h)


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="oracle/_ref/facade_driver_ref not built (needs /root/reference)")
def test_facade_constructs_every_mono_init_extractor_emulated(tmp_path, emu_lib):
    _facade_all(tmp_path, _facade_exe(tmp_path, os.path.join(ROOT, "tests", "emu"), "orbx_emu"), dict(os.environ, ORBX_EMU_LDS_LIMIT=GFX950_LDS))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="oracle/_ref/facade_driver_ref not built")
def test_facade_constructs_every_mono_init_extractor_gpu(tmp_path, hip_lib):
    _facade_all(tmp_path, _facade_exe(tmp_path, os.path.dirname(_lib.HIP_LIB_PATH), "orbx_hip"), dict(os.environ))


def _beyond_4000(lib, w, h, nf, factory):
    """more than 4 000 nodes on a level (12-bit sort ranges of the LDS form) / close to the 65 535 keypoints per image of the 16-bit node indices"""
    img = factory()
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)
    got = ex(img)
    assert ex.debug_quadtree_pool_levels() >= 1 and len(got[1]) > nf * 0.9
    assert _same(got, _expected(nf, 1.2, 8, 20, 7, img, (0, 0)))


def test_pool_form_takes_30000_features_emulated(emu_lib, monkeypatch):
    monkeypatch.setenv("ORBX_EMU_LDS_LIMIT", GFX950_LDS)
    _beyond_4000(emu_lib, 1600, 1200, 30000, lambda: synth.uniform_noise(1600, 1200, seed=23))


@pytest.mark.gpu
def test_pool_form_takes_64000_features_gpu(hip_lib):
    _beyond_4000(hip_lib, 1920, 1200, 64000, lambda: synth.uniform_noise(1920, 1200, seed=22))
    with pytest.raises(_lib.OrbxError, match="nfeatures too large"):
        ORBextractor(66000, 1.2, 8, 20, 7, lib=hip_lib)(synth.sparse_corners(752, 480, seed=4, ncorner=60))
