import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """The kernel sources compiled for the CPU SIMT emulator (tests only)."""
    import subprocess
    from orb_slam3_detailed_comments_amd import _lib
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-C", d, "-s"], check=True, stdout=subprocess.DEVNULL)
    return _lib.OrbxLib(os.path.join(d, "liborbx_emu.so"))


@pytest.fixture(scope="session")
def hip_lib():
    from orb_slam3_detailed_comments_amd import _lib
    lib = _lib.load_hip()
    if lib.L.orbx_device_count() < 1:
        pytest.fail("liborbx_hip.so loaded but no GPU is visible: the gpu tests need an MI355X (no CPU fallback)")
    return lib
