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


# ---- the order of a GPU session ---------------------------------------------------------------------------------
# `pytest -m gpu -x` stops at the first failure: in round 5 one assertion about thread timing stopped the driver's run
# before test_gpu_slab.py and test_gpu_sharded.py had been reached.  So the suite is ordered by what a failure would
# hide: the golden scripts and the oracle-bitwise parity tests first, the full-size configurations next, the slab /
# ingest / shard tests after them, and everything that starts threads or child processes -- where a box's timing can
# show at all -- last.  (Within a group the order stays the files' own.)
_GROUP_OF_FILE = {   # (group, place within it)
    "test_gpu_parity.py": (0, 0), "test_gpu_nif_exec.py": (0, 1), "test_gpu_multiquery.py": (0, 2), "test_gpu_shadow.py": (0, 3),
    "test_gpu_fullsize.py": (1, 0), "test_gpu_config4.py": (1, 1),
    "test_gpu_slab.py": (2, 0), "test_gpu_sharded.py": (2, 1), "test_gpu_ingest.py": (2, 2), "test_gpu_multishard.py": (2, 3),
    "test_gpu_coalesce.py": (3, 0), "test_gpu_bench_supervisor.py": (3, 9), "test_gpu_perf.py": (4, 0),
}
_THREAD_MARKS = ("threading.", "callers_meet(", "_hammer(", "ThreadPoolExecutor")


def _gpu_group(item):
    import inspect
    name = os.path.basename(str(item.fspath))
    group, place = _GROUP_OF_FILE.get(name, (2, 9))
    if group < 3:
        try:
            src = inspect.getsource(item.function)
        except (OSError, TypeError, AttributeError):
            src = ""
        # (test_gpu_sharded.py's two processes and the hooks re-runs are children, not threads: they keep their group)
        if any(m in src for m in _THREAD_MARKS):
            group, place = 3, 1 + place
    return group, place


def pytest_collection_modifyitems(config, items):
    gpu_items = [it for it in items if "gpu" in it.keywords or "gpu_perf" in it.keywords]
    if gpu_items:
        order = {id(it): _gpu_group(it) + (n,) for n, it in enumerate(items) if it in gpu_items}
        rest = [it for it in items if id(it) not in order]
        items[:] = rest + sorted(gpu_items, key=lambda it: order[id(it)])
    # gpu_perf tests run only when asked for by name (-m gpu_perf): `-m "not gpu"` on a CPU box
    # and `-m gpu` on the driver's box both leave them out
    markexpr = config.getoption("-m", default="") or ""
    if "gpu_perf" in markexpr:
        return
    skip = pytest.mark.skip(reason="timing guard: run with -m gpu_perf on an MI355X")
    for item in items:
        if "gpu_perf" in item.keywords:
            item.add_marker(skip)
