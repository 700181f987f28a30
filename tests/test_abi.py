"""The C-ABI shared library (the product, hipcc-built for gfx950) loads and exports every symbol that
include/orbx.h declares; without a GPU it refuses to create an extractor (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import oracle_lib as ol
from orb_slam3_detailed_comments_amd import _lib

ROOT = ol.ROOT


def _ensure_built():
    if not os.path.exists(_lib.HIP_LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "orb_slam3_detailed_comments_amd", "csrc"), "-s"], check=True)


def test_header_symbols_exported():
    _ensure_built()
    hdr = open(os.path.join(ROOT, "include", "orbx.h")).read()
    declared = set(re.findall(r"\b(orb[xmv]_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = C.CDLL(_lib.HIP_LIB_PATH)
    for s in sorted(declared):
        assert hasattr(L, s), s


def test_gfx950_code_object_present():
    _ensure_built()
    data = open(_lib.HIP_LIB_PATH, "rb").read()
    assert b"gfx950" in data
    for k in (b"k_fast_cells", b"k_quadtree", b"k_orient_brief", b"k_stereo_match", b"k_knn2", b"k_resize", b"k_blur"):
        assert k in data, k


def test_no_gpu_means_loud_failure():
    _ensure_built()
    lib = _lib.load_hip()
    if lib.L.orbx_device_count() > 0:
        pytest.skip("a GPU is visible here")
    from orb_slam3_detailed_comments_amd import ORBextractor, OrbxError
    with pytest.raises(OrbxError) as e:
        ORBextractor(1000, 1.2, 8, 20, 7)
    assert e.value.code == _lib.ORBX_E_DEVICE


def test_package_never_references_oracle_or_emulator():
    pkg = os.path.join(ROOT, "orb_slam3_detailed_comments_amd")
    for dp, dn, fn in os.walk(pkg):
        for f in fn:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "liborb_oracle" not in txt and "libref_orb" not in txt, f
                if f.endswith(".py"):
                    assert "liborbx_emu" not in txt or f == "_lib.py", f


def test_header_is_plain_c(tmp_path):
    """include/orbx.h is the C ABI: it must compile as C99 with warnings as errors (no C++ types, no torch types)."""
    src = tmp_path / "cabi.c"
    src.write_text('#include "orbx.h"\nint main(void) { OrbxInputSpec s; OrbmFrameView f; OrbmFisheyeFrameView g; (void)s; (void)f; (void)g; return orbx_device_count() < 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "cabi.o")], check=True)


def test_null_arguments_are_refused_not_dereferenced(emu_lib):
    """~280 calls with null pointers and zero sizes (tests/null_argument_runner.py), on the emulator build of the same host code."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "null_argument_runner.py"), os.path.join(ROOT, "tests", "emu", "liborbx_emu.so"), ROOT],
                       capture_output=True, text=True)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines and lines[-1].startswith("DONE"), "crashed in: %s (rc %d)\n%s" % (lines[-1] if lines else "?", r.returncode, r.stderr[-500:])
    assert int(lines[-1].split()[1]) > 250
