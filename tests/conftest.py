import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a gfx950 GPU (run on the MI355X box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.lib()   # builds liboracle.so on first use (g++, a second or two)
    return oracle


@pytest.fixture(scope="session")
def bunny_scene():
    from raytracer_amd import scenes
    return scenes.bunny()


def random_rays(n, seed, center=(0.0, 0.8, 0.0), spread=6.0):
    """Rays from a box around the scene toward a smaller box at its centre, plus axis-parallel ones."""
    rng = np.random.RandomState(seed)
    c = np.asarray(center, np.float32)
    org = ((rng.rand(n, 3).astype(np.float32) - 0.5) * spread + c + np.array([0, 0.7, 0], np.float32)).astype(np.float32)
    tgt = ((rng.rand(n, 3).astype(np.float32) - 0.5) * 2 + c).astype(np.float32)
    d = tgt - org
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    k = max(1, n // 40)
    d[:k] = np.array([0, -1, 0], np.float32)      # straight down: two zero components
    d[k:2 * k] = np.array([1, 0, 0], np.float32)
    d[2 * k:3 * k, 0] = 0.0                        # one zero component (not renormalised on purpose)
    return org, d


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
