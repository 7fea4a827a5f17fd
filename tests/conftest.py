import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the library reads its test hooks (KATGPU_TEST_*, the A/B switches) only when this is set; subprocesses inherit it
os.environ.setdefault("KATGPU_TESTING", "1")

REFDATA = os.path.join(ROOT, "tests", "golden", "refdata")   # data files the reference's own tests hold (tests/data/)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ko():
    """The CPU oracle (oracle/ -- test infrastructure)."""
    from oracle import koracle
    koracle.lib()
    return koracle


@pytest.fixture(scope="session")
def engine():
    """One katgpu context on device 0.  Fails loudly (no skip, no fallback) if the HIP library or the GPU is missing."""
    import kat_amd
    eng = kat_amd.Engine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def refdata():
    return REFDATA
