import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    # torch bundles its own libamdhip64 (same soname as ROCm's): whichever copy is
    # loaded first serves the whole process, and torch does not survive coming
    # second.  GPU sessions use both, so torch goes first.
    markexpr = config.getoption("-m", default="") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
