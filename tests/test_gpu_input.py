"""Input pre-step on the real GPU at EuRoC size (752x480: rectification, 600x350 resize, colour frames)."""
import pytest

from test_emu_input import run

pytestmark = pytest.mark.gpu


def test_input_prestep_gpu(hip_lib):
    run(None, 752, 480, 1200)
