"""The host side's thread machinery -- the coalescer, the context lease, the per-shard workers
(vettore_amd/csrc/host/vt_concurrency.h, stand-alone templates the product instantiates with the
real operations) -- built here with stub operations and run under ThreadSanitizer: 64 callers x
mixed limits x injected failures x a writer forcing the disband path, 10^5 searches, zero reports,
nobody left waiting (two races in this code were found in round 2 by a test
that failed one run in three)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_concurrency_under_thread_sanitizer():
    exe = os.path.join(tempfile.mkdtemp(), "concurrency_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsanitize=thread", "-pthread",
                           os.path.join(ROOT, "tests", "concurrency_check.cpp"), "-o", exe])
    # (VT_COALESCE_SLOTS=3: what the settings table must still say after a thread has spent the run calling setenv)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66", VT_COALESCE_SLOTS="3")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and out.stdout.strip() == "ok" and "ThreadSanitizer" not in out.stderr, \
        (out.returncode, out.stdout, out.stderr[-3000:])
    assert "failed batches" in out.stderr and "beside setenv" in out.stderr and "second contexts handed out" in out.stderr


def test_the_product_uses_these_templates():
    """The check above is only worth something while the library runs the same code."""
    types = open(os.path.join(ROOT, "vettore_amd", "csrc", "host", "vt_types.h")).read()
    coal = open(os.path.join(ROOT, "vettore_amd", "csrc", "host", "vt_coalesce.h")).read()
    multi = open(os.path.join(ROOT, "vettore_amd", "csrc", "host", "vt_multi.h")).read()
    assert "vt_host::LeaseT<Shard, Ctx>" in types and "vt_host::WorkerT<HipWorkerPolicy>" in types
    assert "vt_host::coalesced_search_t<vt_flat, CoalesceOps>" in coal and "vt_host::run_coalesced_t<vt_flat, CoalesceOps>" in coal
    assert "vt_host::run_on_workers" in multi and "vt_host::SpareLeaseT<Shard, Ctx>" in types
    # ... and reads its switches from the table checked above: the environment once, at load
    import glob
    import re
    host = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "vettore_amd", "csrc", "host", "*.h")))
    assert not re.search(r"\bgetenv\s*\(", host)
    env_h = open(os.path.join(ROOT, "vettore_amd", "csrc", "vt_env.h")).read()
    assert len(re.findall(r"\bgetenv\s*\(", env_h)) == 1
    for src in glob.glob(os.path.join(ROOT, "vettore_amd", "csrc", "*.hip")) + [os.path.join(ROOT, "vettore_amd", "csrc", "vt_index.cpp")]:
        assert not re.search(r"\bgetenv\s*\(", open(src).read()), src
