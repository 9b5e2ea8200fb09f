"""Shared fixtures.  `-m gpu` tests need a real MI355X; everything else runs on CPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fldr-vfi_amd")
for p in (PKG, os.path.join(ROOT, "oracle")):       # oracle: tests are allowed to import it
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
WEIGHTS = os.path.join(PKG, "weights", "fLDRnet_X4K1000FPS_exp1_best_PSNR.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import fldr_oracle
    return fldr_oracle


@pytest.fixture(scope="session")
def weights(oracle):
    return oracle.load_weights(WEIGHTS)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _product_ring_never_timed_out():
    """After the session: the PRODUCT library's convolution ring never ran on past an expired wait (fldr_range_status bit 1) —
    every model-level test above ran on that library, so a silent wrong-output convolution would show here."""
    yield
    if torch.cuda.is_available() and "fldr_hip" in sys.modules:
        import fldr_hip
        assert not (fldr_hip.device_status(reset=False) & fldr_hip.STATUS_RING_TIMEOUT), "a ring wait expired in libfldr_hip.so"
