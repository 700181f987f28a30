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


@pytest.mark.parametrize("presort_max,bigspan", [(1, 80), (0, 1024)])
def test_quadtree_path_mix(tmp_path, presort_max, bigspan):
    so = str(tmp_path / "liborbx_emu_variant.so")
    srcs = [os.path.join(CSRC, f) for f in ("k_image.hip", "k_quadtree.hip", "k_describe.hip", "k_match.hip", "k_search.hip", "k_vocab.hip", "k_input.hip", "orbx_api.cpp", "orbm_search.cpp", "orbv_api.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-DORBX_EMU", "-DORBX_PRESORT_MAX=%d" % presort_max, "-DORBX_BIGSPAN=%d" % bigspan,
                    "-I" + os.path.join(ROOT, "tests", "emu"), "-I" + CSRC, "-fPIC", "-shared", "-w", "-x", "c++"] + srcs + ["-o", so, "-lpthread"], check=True)
    lib = _lib.OrbxLib(so)
    for name, factory, nf, lap in SMALL_CASES + FULL_CASES[:1] + FULL_CASES[4:5]:
        img = factory()
        got = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)(img, None, lap)
        exp = ol.OracleExtractor(nf).extract(img, lap)
        assert got[0] == exp[0] and ol.kps_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]), name


def test_fast_list_flush_path(tmp_path):
    """k_fast_cells keeps survivors of the quick test in a list sized for half of a cell's pixels and flushes it when a denser cell
    would overflow; with a 520-entry list nearly every cell of a textured image takes that path (and its list-free NMS / compaction)."""
    so = str(tmp_path / "liborbx_emu_smalllist.so")
    srcs = [os.path.join(CSRC, f) for f in ("k_image.hip", "k_quadtree.hip", "k_describe.hip", "k_match.hip", "k_search.hip", "k_vocab.hip", "k_input.hip", "orbx_api.cpp", "orbm_search.cpp", "orbv_api.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-DORBX_EMU", "-DORBX_FAST_LIST_CAP=520",
                    "-I" + os.path.join(ROOT, "tests", "emu"), "-I" + CSRC, "-fPIC", "-shared", "-w", "-x", "c++"] + srcs + ["-o", so, "-lpthread"], check=True)
    lib = _lib.OrbxLib(so)
    for name, factory, nf, lap in SMALL_CASES[:3] + FULL_CASES[:1] + FULL_CASES[4:5]:
        img = factory()
        got = ORBextractor(nf, 1.2, 8, 20, 7, lib=lib)(img, None, lap)
        exp = ol.OracleExtractor(nf).extract(img, lap)
        assert got[0] == exp[0] and ol.kps_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]), name
