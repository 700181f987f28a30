"""Guided searches on the real GPU at BASELINE.json config-4 sizes (640x480, nFeatures=1000, 5000 map points)."""
import numpy as np
import pytest

from test_emu_search import run_all

pytestmark = pytest.mark.gpu


def test_guided_searches_gpu_full_size(hip_lib):
    run_all(None, 640, 480, 1000, 5000, seeds=(0, 1, 2))


def test_guided_searches_gpu_euroc_size(hip_lib):
    run_all(None, 752, 480, 1200, 3000, seeds=(5,))


def test_distinctive_descriptors_gpu(hip_lib):
    from test_emu_mappoint import run
    run(None, 5000, 60, 2)          # config-4 scale: ~5000 local map points
    run(None, 200, 400, 3)


def test_matchers_from_three_threads_gpu(hip_lib):
    from test_emu_search import concurrent_matchers
    concurrent_matchers(None, 640, 480, 1000, 3000)
