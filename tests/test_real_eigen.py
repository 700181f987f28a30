"""Independent check of the stand-in Eigen / Sophus of oracle/slam_shim against a REAL Eigen ("parity unpinned" for
the restated evaluation orders).  Eigen is not installed in the build image, so the test SKIPS there; wherever /usr/include/eigen3 (or $EIGEN3_INCLUDE)
exists and the reference's vendored Sophus is present, tests/cpp/eigen_probe.cpp is compiled twice - against real Eigen + Sophus, and against the
stand-ins - with -ffp-contract=off, and the bit patterns of 2000 seeded rounds of dot / norm / 3x3 products and inverses / SE3 actions on unit
quaternions / JacobiSVD<Matrix4f> null vectors must be identical."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EIGEN = os.environ.get("EIGEN3_INCLUDE", "/usr/include/eigen3")
REF = "/root/reference"
SOPHUS = os.path.join(REF, "Thirdparty", "Sophus")
SRC = os.path.join(ROOT, "tests", "cpp", "eigen_probe.cpp")
SHIM = ["-I" + os.path.join(ROOT, "oracle", d) for d in ("opencv_shim", "boost_shim", "slam_shim")] + ["-I" + REF, "-I" + os.path.join(REF, "Thirdparty", "DBoW2"),
                                                                                                         "-I" + os.path.join(REF, "include"), "-I" + os.path.join(REF, "include", "CameraModels")]
FLAGS = ["-std=c++14", "-O2", "-march=x86-64-v2", "-ffp-contract=off", "-w"]


def _run(tmp_path, name, extra):
    exe = str(tmp_path / name)
    subprocess.run(["g++"] + FLAGS + extra + [SRC, "-o", exe], check=True)
    return subprocess.run([exe], capture_output=True, text=True, check=True).stdout


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "include", "CameraModels", "Pinhole.h")), reason="needs /root/reference (the stand-in world includes its camera headers)")
def test_stand_in_probe_runs(tmp_path):
    """the stand-in build of the probe compiles and produces its 2000 rounds (so that the comparison below cannot rot while it is skipped)"""
    out = _run(tmp_path, "probe_shim", SHIM + ["-include", os.path.join(ROOT, "oracle", "slam_shim", "slam_world.h")])
    assert out.count("\n") == 2000 * (2 + 3 + 3 + 9 + 9 + 3 + 3 + 9 + 3 + 4) and "nan" not in out


HAVE_REAL = os.path.exists(os.path.join(EIGEN, "Eigen", "Dense")) and os.path.exists(os.path.join(SOPHUS, "sophus", "se3.hpp"))


# A CI that is supposed to carry a real Eigen sets ORBX_REQUIRE_REAL_EIGEN=1: the comparison then FAILS instead of skipping when the headers are missing
# (a skipped pin is no pin).  The build image has no Eigen anywhere (no network): there it skips, and DESIGN.md section 2.2 says "unpinned".
@pytest.mark.skipif(not HAVE_REAL and os.environ.get("ORBX_REQUIRE_REAL_EIGEN") != "1",
                    reason="no real Eigen on this machine (install libeigen3-dev or set EIGEN3_INCLUDE) / no vendored Sophus")
def test_stand_in_equals_real_eigen(tmp_path):
    assert HAVE_REAL, "ORBX_REQUIRE_REAL_EIGEN=1 but %s/Eigen/Dense or %s/sophus/se3.hpp is missing" % (EIGEN, SOPHUS)
    real = _run(tmp_path, "probe_real", ["-DPROBE_REAL_EIGEN", "-I" + EIGEN, "-I" + SOPHUS])
    shim = _run(tmp_path, "probe_shim", SHIM + ["-include", os.path.join(ROOT, "oracle", "slam_shim", "slam_world.h")])
    a, b = real.splitlines(), shim.splitlines()
    assert len(a) == len(b)
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
    assert not diff, "%d of %d values differ from real Eigen / Sophus, first: %r" % (len(diff), len(a), diff[:5])
