import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENE = os.path.join(ROOT, "lego_rust")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def samples(oracle):
    return oracle.load_samples(os.path.join(SCENE, "tf_reference_samples.json"))


@pytest.fixture(scope="session")
def oracle_nets(oracle):
    return oracle.Net(os.path.join(SCENE, "coarse")), oracle.Net(os.path.join(SCENE, "fine"))


@pytest.fixture(scope="session")
def native():
    """The product package over libnerf_mi355x.so (must exist: there is no fallback)."""
    import nerf_rs_amd
    nerf_rs_amd.load_library()
    return nerf_rs_amd


@pytest.fixture(scope="session")
def renderer(native):
    r = native.Renderer(0)
    r.load_scene(SCENE)
    yield r
    r.close()


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def psnr(a, b):
    mse = float(np.mean((np.clip(a, 0, 1).astype(np.float64) - np.clip(b, 0, 1).astype(np.float64)) ** 2))
    return 99.0 if mse == 0 else 10.0 * np.log10(1.0 / mse)
