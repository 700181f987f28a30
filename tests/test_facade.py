"""Drop-in check of the C++ facade: tests/cpp/facade_driver.cpp is written against the REFERENCE's public ORBextractor
interface only.  It is built once against the reference's own header + source (oracle/_ref/facade_driver_ref) and once
against include/orb_slam3_amd/ORBextractor.h + the product library; the two binary dumps (keypoints, descriptors, the public
mvImagePyramid incl. its 19-px borders, the scale getters) must be identical."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from orb_slam3_detailed_comments_amd import synth, _lib

ROOT = ol.ROOT
REF = os.path.join(ROOT, "oracle", "_ref", "facade_driver_ref")


def _run_pair(tmp_path, libdir, libname, extra_env=None):
    img = synth.corner_field(376, 240, seed=10, nrect=800)
    raw = tmp_path / "im.raw"; raw.write_bytes(img.tobytes())
    exe = tmp_path / "facade_driver_ours"
    subprocess.run(["g++", "-std=c++14", "-O1", "-w", "-I" + os.path.join(ROOT, "include", "orb_slam3_amd"), "-I" + os.path.join(ROOT, "oracle", "opencv_shim"),
                    os.path.join(ROOT, "tests", "cpp", "facade_driver.cpp"), "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    for lap in ((0, 0), (100, 250)):
        a, b = tmp_path / "ref.bin", tmp_path / "ours.bin"
        args = [str(raw), "376", "240", "500", str(lap[0]), str(lap[1])]
        subprocess.run([REF] + args + [str(a)], check=True)
        env = dict(os.environ); env.update(extra_env or {})
        subprocess.run([str(exe)] + args + [str(b)], check=True, env=env)
        assert a.read_bytes() == b.read_bytes(), "facade output differs from the reference build (lap=%s)" % (lap,)
        assert a.stat().st_size > 100000


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/facade_driver_ref not built (needs /root/reference)")
def test_facade_dropin_emulated(tmp_path, emu_lib):
    _run_pair(tmp_path, os.path.join(ROOT, "tests", "emu"), "orbx_emu")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/facade_driver_ref not built")
def test_facade_dropin_gpu(tmp_path, hip_lib):
    _run_pair(tmp_path, os.path.dirname(_lib.HIP_LIB_PATH), "orbx_hip")


def _run_matcher_facade(tmp_path, libdir, libname):
    L, R = synth.stereo_pair(376, 240, seed=20, nrect=800)
    (tmp_path / "l.raw").write_bytes(L.tobytes()); (tmp_path / "r.raw").write_bytes(R.tobytes())
    ol.oracle()
    exe = tmp_path / "matcher_facade_test"
    subprocess.run(["g++", "-std=c++14", "-O1", "-w", "-I" + os.path.join(ROOT, "include", "orb_slam3_amd"), "-I" + os.path.join(ROOT, "oracle", "opencv_shim"),
                    os.path.join(ROOT, "tests", "cpp", "matcher_facade_test.cpp"), "-L" + libdir, "-l" + libname, "-L" + os.path.join(ROOT, "oracle"), "-lorb_oracle",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), "376", "240"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _run_matcher_threads(tmp_path, libdir, libname):
    """Three threads call the ORBmatcher facade concurrently (one library handle per thread): the same results as one after the other."""
    L, _ = synth.stereo_pair(376, 240, seed=21, nrect=800)
    (tmp_path / "t.raw").write_bytes(L.tobytes())
    exe = tmp_path / "matcher_threads_test"
    subprocess.run(["g++", "-std=c++14", "-O1", "-w", "-I" + os.path.join(ROOT, "include", "orb_slam3_amd"), "-I" + os.path.join(ROOT, "oracle", "opencv_shim"),
                    os.path.join(ROOT, "tests", "cpp", "matcher_threads_test.cpp"), "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-lpthread", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), str(tmp_path / "t.raw"), "376", "240"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return exe


def test_matcher_facade_three_threads_emulated(tmp_path, emu_lib):
    exe = _run_matcher_threads(tmp_path, os.path.join(ROOT, "tests", "emu"), "orbx_emu")
    # a multi-GPU C++ host: two "GPUs" (the emulator reports as many as ORBX_EMU_DEVICES says), extractor + matcher handles per device, and the
    # descriptor all-gather of BASELINE.json configs[4] through the C ABI alone (orbx_comm_*, orbx_allgather_descriptors)
    r = subprocess.run([str(exe), str(tmp_path / "t.raw"), "376", "240", "2"], capture_output=True, text=True, timeout=600, env=dict(os.environ, ORBX_EMU_DEVICES="2"))
    assert r.returncode == 0 and "ranks=2 exchange mismatches=0" in r.stdout, r.stdout + r.stderr
    # ... and a device the node does not have is refused, not silently mapped to GPU 0
    r = subprocess.run([str(exe), str(tmp_path / "t.raw"), "376", "240", "2"], capture_output=True, text=True, timeout=600, env=dict(os.environ, ORBX_EMU_DEVICES="1"))
    assert r.returncode != 0


@pytest.mark.gpu
def test_matcher_facade_three_threads_gpu(tmp_path, hip_lib):
    _run_matcher_threads(tmp_path, os.path.dirname(_lib.HIP_LIB_PATH), "orbx_hip")


def test_matcher_facade_emulated(tmp_path, emu_lib):
    _run_matcher_facade(tmp_path, os.path.join(ROOT, "tests", "emu"), "orbx_emu")


@pytest.mark.gpu
def test_matcher_facade_gpu(tmp_path, hip_lib):
    _run_matcher_facade(tmp_path, os.path.dirname(_lib.HIP_LIB_PATH), "orbx_hip")


DBOW2 = "/root/reference/Thirdparty/DBoW2"


@pytest.mark.skipif(not os.path.isdir(DBOW2), reason="needs the reference's DBoW2 sources (/root/reference)")
def test_vocabulary_facade_emulated(tmp_path, emu_lib):
    """include/orb_slam3_amd/ORBVocabulary.h vs the reference's DBoW2 in one binary: equal BowVector / FeatureVector objects."""
    import vocab_scenes as vs
    rng = np.random.default_rng(2)
    header, parent, leaf, desc, weight = vs.make_vocabulary(rng, 10, 3)
    voc = tmp_path / "voc.txt"
    vs.write_text(voc, header, parent, leaf, desc, weight)
    exe = tmp_path / "vocabulary_facade_test"
    libdir = os.path.join(ROOT, "tests", "emu")
    srcs = [os.path.join(DBOW2, "DBoW2", f) for f in ("FORB.cpp", "BowVector.cpp", "FeatureVector.cpp", "ScoringObject.cpp")] + [os.path.join(DBOW2, "DUtils", "Random.cpp"), os.path.join(DBOW2, "DUtils", "Timestamp.cpp")]
    subprocess.run(["g++", "-std=c++14", "-O1", "-w", "-I" + os.path.join(ROOT, "include", "orb_slam3_amd"), "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "oracle", "opencv_shim"), "-I" + os.path.join(ROOT, "oracle", "boost_shim"), "-I" + DBOW2,
                    os.path.join(ROOT, "tests", "cpp", "vocabulary_facade_test.cpp")] + srcs +
                   ["-L" + libdir, "-lorbx_emu", "-Wl,-rpath," + libdir, "-lpthread", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), str(voc), "900"], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_facade_headers_link_from_several_translation_units(tmp_path, emu_lib):
    """Tracking.cc, Frame.cc, MapPoint.cc, LocalMapping.cc ... all include ORBmatcher.h / ORBextractor.h: the drop-in headers must not define
    anything that collides at link time (they once defined TH_LOW / TH_HIGH / HISTO_LENGTH at namespace scope)."""
    voc = os.path.isdir(DBOW2)
    inc = '#include "ORBextractor.h"\n#include "ORBmatcher.h"\n' + ('#include "ORBVocabulary.h"\n' if voc else "")
    (tmp_path / "a.cpp").write_text(inc + "int fa() { ORB_SLAM3::ORBmatcher m(0.7f, true); const int& r = ORB_SLAM3::ORBmatcher::TH_LOW; return r; }\n")
    (tmp_path / "b.cpp").write_text(inc + "int fa();\nint main() { const int& r = ORB_SLAM3::ORBmatcher::TH_HIGH; return fa() + r + ORB_SLAM3::ORBmatcher::HISTO_LENGTH == 180 ? 0 : 1; }\n")
    libdir = os.path.join(ROOT, "tests", "emu")
    cmd = ["g++", "-std=c++14", "-w", "-I" + os.path.join(ROOT, "include", "orb_slam3_amd"), "-I" + os.path.join(ROOT, "oracle", "opencv_shim")]
    if voc:
        cmd += ["-I" + DBOW2, "-I/root/reference", "-I" + os.path.join(ROOT, "oracle", "boost_shim")]
    exe = tmp_path / "two_tu"
    subprocess.run(cmd + [str(tmp_path / "a.cpp"), str(tmp_path / "b.cpp"), "-L" + libdir, "-lorbx_emu", "-Wl,-rpath," + libdir, "-lpthread", "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
