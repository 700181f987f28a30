"""The quadtree kernel has three code paths for dividing a node (bucket offsets of the up-front counting sort, a one-wave
register/loop partition, a workgroup-cooperative partition).  With the production thresholds small test images mostly take the
first one, so the kernel sources are also built with a shallow presort and a low cooperative threshold to force every path."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from cases import SMALL_CASES, FULL_CASES
from orb_slam3_detailed_comments_amd import _lib
from orb_slam3_detailed_comments_amd.extractor import ORBextractor

ROOT = ol.ROOT
CSRC = os.path.join(ROOT, "orb_slam3_detailed_comments_amd", "csrc")
SOURCES = ("k_image.hip", "k_fast.hip", "k_quadtree.hip", "k_describe.hip", "k_match.hip", "k_search.hip", "k_vocab.hip", "k_input.hip", "orbx_api.cpp", "orbm_search.cpp",
           "orbv_api.cpp", "orbx_comm.cpp")


def build_emu_variant(so, defines):
    """the kernel sources for the CPU SIMT emulator with extra -D switches; the translation units are compiled side by side"""
    from concurrent.futures import ThreadPoolExecutor
    d = os.path.dirname(so)
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-fwrapv", "-fno-gnu-unique", "-DORBX_EMU"] + list(defines) + ["-I" + os.path.join(ROOT, "tests", "emu"), "-I" + CSRC, "-fPIC", "-w"]

    def one(f):
        o = os.path.join(d, f + ".o")
        subprocess.run(["g++"] + flags + ["-c", "-x", "c++", os.path.join(CSRC, f), "-o", o], check=True)
        return o
    with ThreadPoolExecutor(6) as ex:
        objs = list(ex.map(one, SOURCES))
    subprocess.run(["g++", "-shared"] + objs + ["-o", so, "-lpthread"], check=True)


@pytest.mark.parametrize("presort_max,bigspan,wave_sort_range,big_pixels,u16_max", [(1, 80, 16, 0, 65535), (0, 1024, 100000, 150000, 0), (5, 1024, 40, 0, 0)])
def test_quadtree_path_mix(tmp_path, presort_max, bigspan, wave_sort_range, big_pixels, u16_max):
    # big_pixels: levels with at least this many pixels run the quadtree with 1024 threads (0 = every level);
    # the second build also never widens (every level on 256 threads, as large batches run); u16_max: levels with more keys
    # count their buckets in 32-bit counters and fewer segments (0 = every level);
    # the second build also never widens (every level on 256 threads, as large batches run)
    # wave_sort_range: ranges of the final rounds' std::sort model above this length are partitioned by a wave (16 = every range,
    # 100000 = none: the one-thread loop)
    so = str(tmp_path / "liborbx_emu_variant.so")
    build_emu_variant(so, ["-DORBX_PRESORT_MAX=%d" % presort_max, "-DORBX_BIGSPAN=%d" % bigspan, "-DORBX_WAVE_SORT_RANGE=%d" % wave_sort_range, "-DORBX_QT_BIG_PIXELS=%d" % big_pixels,
                           "-DORBX_PRESORT_U16_MAX=%d" % u16_max, "-DORBX_QT_WIDE_BATCH=%d" % (0 if big_pixels == 150000 else 32)])
    lib = _lib.OrbxLib(so)
    for name, factory, nf, lap in SMALL_CASES + FULL_CASES[:1] + FULL_CASES[4:5]:
        img = factory()
        got = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)(img, None, lap)
        exp = ol.OracleExtractor(nf).extract(img, lap)
        assert got[0] == exp[0] and ol.kps_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]), name


@pytest.mark.parametrize("cap", [200, 600])
def test_fast_list_flush_path(tmp_path, cap):
    """k_fast_cells keeps the corners found so far and the survivors of the quick test that wait for their score in one list; when it fills
    up the pending ones are scored, and a cell whose corners alone fill it (noise) falls back to NMS / compaction scans of the score tile.
    With 600 entries most cells of a textured image score in several batches, with 200 nearly all of them also give the corner list up."""
    so = str(tmp_path / "liborbx_emu_smalllist.so")
    build_emu_variant(so, ["-DORBX_FAST_LIST_CAP=%d" % cap])
    lib = _lib.OrbxLib(so)
    # low-amplitude noise: next to no corner at iniThFAST, a crowd at minThFAST - the second FAST run of a cell takes the flush / list-free paths
    low_noise = ("low_noise", lambda: (60 + np.random.default_rng(5).integers(0, 26, (240, 376))).astype(np.uint8), 500, (0, 0))
    fallback = [c for c in SMALL_CASES if c[0] == "small_threshold_fallback"]
    assert fallback
    for name, factory, nf, lap in SMALL_CASES[:3] + FULL_CASES[:1] + FULL_CASES[4:5] + [low_noise] + fallback:
        img = factory()
        got = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)(img, None, lap)
        exp = ol.OracleExtractor(nf).extract(img, lap)
        assert got[0] == exp[0] and ol.kps_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]), name


@pytest.mark.parametrize("strip", [32, 64])
def test_resize_strips_longer_than_a_pitch_quantum(tmp_path, strip):
    """k_resize_rows: lane i of a wave carries the vertical taps of output row ys + i for the whole wave.  Until round 6 the lanes beyond the row pitch left
    the kernel at once - correct only while a strip had at most 16 rows (a 64-byte pitch always fills 16 lanes); a build with ORBX_RESIZE_STRIP = 32 read
    taps out of dead lanes on the last workgroup of a row (found when the strip length was tried as an experiment: parity_check of bench.py failed).  Now every
    lane stays alive (the surplus lanes redo the row's last dword).  Batches above 32 images use the configured strip length."""
    from orb_slam3_detailed_comments_amd import synth
    so = str(tmp_path / "liborbx_emu_strip.so")
    build_emu_variant(so, ["-DORBX_RESIZE_STRIP=%d" % strip, "-DORBX_RESIZE_MIN_BLOCKS=1"])       # (the library then keeps the configured strip at any batch size)
    lib = _lib.OrbxLib(so)
    for (w, h) in ((522, 333), (752, 480)):                        # level widths 435, 363, 302 ..: pitches whose last workgroup fills 48, 32, 16 .. lanes
        img = synth.pink_noise(w, h, seed=w)
        other = synth.sparse_corners(w, h, seed=5, ncorner=20)
        o = ol.OracleExtractor(300); o.extract(img)
        ex = ORBextractor(300, 1.2, 8, 20, 7, lib=lib)
        ex.extract_batch(np.stack([other] * 33 + [img]))           # 34 images: the large-batch launch form (one launch per level, configured strips)
        for l in range(8):
            assert np.array_equal(ex.pyramid_level(l, 33), o.level_image(l)), "level %d of %dx%d, strips of %d rows" % (l, w, h, strip)
