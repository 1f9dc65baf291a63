import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "gpu_perf: timing guards and throughput floors on a real MI355X "
                                       "(run with -m gpu_perf; kept out of -m gpu so a busy box cannot fail parity)")
    # torch bundles its own libamdhip64 (same soname as ROCm's): whichever copy is
    # loaded first serves the whole process, and torch does not survive coming
    # second.  GPU sessions use both, so torch goes first.
    # (also when GPU test files are named without -m: `pytest tests/test_gpu_x.py`)
    markexpr = config.getoption("-m", default="") or ""
    named = any("test_gpu_" in str(a) for a in config.args)
    if ("gpu" in markexpr or named) and "not gpu" not in markexpr.replace("not gpu_perf", ""):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


class _Settings:
    """Library settings for the length of one test (vt_debug_set, include/vettore_flat.h).  The library reads
    the VT_* variables once, when it is loaded -- long before a test body runs -- so a test names the setting
    ("batch_no_mfma", "coalesce_slots", "force_batch_mfma" ...) instead of patching the environment."""

    def __init__(self):
        self.saved = {}

    def set(self, name, value):
        from vettore_amd import nifs
        if name not in self.saved:
            self.saved[name] = nifs.debug_get(name)
        nifs.debug_set(name, value)

    def reset(self, name):
        from vettore_amd import nifs
        if name in self.saved:
            nifs.debug_set(name, self.saved.pop(name))

    def restore(self):
        for name in list(self.saved):
            self.reset(name)


@pytest.fixture
def vt_debug():
    s = _Settings()
    yield s
    s.restore()


def pytest_collection_modifyitems(config, items):
    # gpu_perf tests run only when asked for by name (-m gpu_perf): `-m "not gpu"` on a CPU box
    # and `-m gpu` on the driver's box both leave them out
    markexpr = config.getoption("-m", default="") or ""
    if "gpu_perf" in markexpr:
        return
    skip = pytest.mark.skip(reason="timing guard: run with -m gpu_perf on an MI355X")
    for item in items:
        if "gpu_perf" in item.keywords:
            item.add_marker(skip)
