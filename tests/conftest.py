import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (first: PyTorch ships its own HIP runtime, and its device initialisation fails in a process
              # that has already loaded the system one through libsrt_hip.so; some GPU tests use torch device buffers)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def scenes(orc):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = orc.load_scene_npz(os.path.join(GOLD, "scenes", name + ".npz"))
        return cache[name]

    return get


@pytest.fixture(scope="session")
def leaf_golden():
    with np.load(os.path.join(GOLD, "leaf_math.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def camera_golden():
    with np.load(os.path.join(GOLD, "camera.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        idx = np.argwhere(~same)[:5]
        raise AssertionError(f"{what}: {np.count_nonzero(~same)} of {same.size} values differ, first at {idx.tolist()}: "
                             f"{a[tuple(idx[0])]!r} vs {b[tuple(idx[0])]!r}")
