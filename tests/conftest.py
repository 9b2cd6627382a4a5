import os
import sys

import numpy as np
import pytest

try:
    # torch ships its own HIP runtime; when tests hand torch device tensors to the engine, torch's copy has to be the
    # first one in the process (loading libddmpc.so first leaves torch with "No HIP GPUs are available")
    import torch  # noqa: F401
except Exception:           # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Fixtures generated from the reference's importable modules (tests/golden/make_golden.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "four_tank_golden.npz"))


def _gpu_count():
    try:
        from direct_data_driven_mpc_amd import _lib
        return _lib.load().ddmpc_device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu():
    """Skip (never silently fall back) when a gpu-marked test is collected without a device."""
    n = _gpu_count()
    if n <= 0:
        pytest.skip("no HIP device visible")
    return n
