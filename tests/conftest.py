import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: oracle pinning, models, the kernel sources on the SIMT emulator) is ~200 independent tests; when pytest-xdist is installed it is spread over the cores
    (a minute and a half on 8 since the emulator switches fibers without system calls and keeps its worker threads, round 5; 7 minutes before).  GPU runs (`-m gpu`) stay in one process: the
    tests time kernels and share one device.  An explicit -n / -p no:xdist on the command line wins."""
    try:
        import xdist  # noqa: F401
    except Exception:
        return None
    opt = config.option
    if getattr(opt, "numprocesses", None) is None and getattr(opt, "markexpr", "") == "not gpu" and not os.environ.get("PYTEST_XDIST_WORKER") \
            and config.pluginmanager.hasplugin("xdist"):
        opt.numprocesses = max(1, min(6, (os.cpu_count() or 2) - 2))
        if opt.numprocesses > 1:
            opt.dist = "load"
            opt.tx = ["popen"] * opt.numprocesses
        else:
            opt.numprocesses = None
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """The kernel sources compiled for the CPU SIMT emulator (tests only)."""
    import subprocess
    from orb_slam3_detailed_comments_amd import _lib
    import fcntl
    d = os.path.join(ROOT, "tests", "emu")
    with open(os.path.join(d, ".build.lock"), "w") as lock:          # the workers of a parallel run build it once, one after the other
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.run(["make", "-C", d, "-s"], check=True, stdout=subprocess.DEVNULL)
    # ORBX_EMU_LIB: another build of the same sources (tools/emu_sanitizers.sh: AddressSanitizer / UBSan builds, run with libasan preloaded)
    return _lib.OrbxLib(os.environ.get("ORBX_EMU_LIB") or os.path.join(d, "liborbx_emu.so"))


@pytest.fixture(scope="session")
def hip_lib():
    from orb_slam3_detailed_comments_amd import _lib
    lib = _lib.load_hip()
    if lib.L.orbx_device_count() < 1:
        pytest.fail("liborbx_hip.so loaded but no GPU is visible: the gpu tests need an MI355X (no CPU fallback)")
    return lib
