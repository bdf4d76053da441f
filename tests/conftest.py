import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


FWD_CASES = ["fwd_sigA_small", "fwd_sigC_demo", "fwd_sigB_y19", "fwd_130_128"]


@pytest.fixture(scope="session")
def oracle_clib():
    """ctypes handle of the plain-C oracle (built by `make -C oracle`; test infrastructure)."""
    import ctypes
    import subprocess
    path = os.path.join(ROOT, "oracle", "libngcf_oracle.so")
    src = os.path.join(ROOT, "oracle", "ngcf_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(path)


@pytest.fixture
def lib_options():
    """Set tunables of libngcf_hip.so by name (ngcf_set_option) for one test; the environment's values come back afterwards."""
    from seoul_tourism_recommendation_ngcf_amd import _lib

    def set_(**kw):
        for k, v in kw.items():
            _lib.set_option(k, v)
    yield set_
    _lib.options_from_env()
