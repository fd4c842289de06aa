import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return load


@pytest.fixture
def lib_option():
    """Scoped msm_set_option overrides (kernel-selection switches of libmsm_hip.so): ``lib_option("MASK_NC", 1)``;
    every option the test touched is restored to MSM_OPT_AUTO afterwards."""
    from unseenobjectswithmeanshift_amd import _lib
    touched = []

    def set_(name, value):
        touched.append(name)
        _lib.set_option(name, value)

    yield set_
    for name in touched:
        _lib.set_option(name, _lib.OPT_AUTO)
