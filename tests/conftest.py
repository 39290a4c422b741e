import os
import sys

import pytest

# PyTorch ships its own copy of the HIP runtime.  In a process that uses both, torch must be imported BEFORE libpie_hip.so
# initialises HIP (the other order leaves torch with "No HIP GPUs are available" on this image); the tests that pass torch
# tensors to the C ABI rely on this import, whatever subset of the test files is collected.
try:
    import torch  # noqa: F401
except Exception:  # torch is only needed by the exchange-step tests
    torch = None

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def pie():
    import sph_pie_amd
    sph_pie_amd.build_hip()
    return sph_pie_amd


@pytest.fixture(scope="session")
def gpu_ctx(pie):
    """One PieScan context for the whole GPU session (one process, one GPU)."""
    ctx = pie.PieScan(0)
    yield ctx
    ctx.close()
