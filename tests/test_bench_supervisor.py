"""bench.py's supervisor on a CPU box: `bench.py --gpus N` runs its measurement in a
child process and, should that child fail or hang, once more over the host exchange -- the first
multi-GPU run of a node must not come back empty.  The child here is tests/stubs/fake_bench_child.py
(VT_BENCH_CHILD); the GPU leg of the same path is tests/test_gpu_bench_supervisor.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
FAKE = os.path.join(ROOT, "tests", "stubs", "fake_bench_child.py")


def run(args, fake, timeout=60, **env):
    e = dict(os.environ, VT_BENCH_CHILD=FAKE, FAKE_CHILD=fake)
    e.update({k: str(v) for k, v in env.items()})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        if k not in env:
            e.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text   # ONE JSON line, whatever the children printed
    return json.loads(lines[0])


def test_a_healthy_rccl_run_is_printed_as_it_is():
    r = run(["--gpus", "2", "--steps", "3"], "ok")
    assert r.returncode == 0, r.stderr
    line = last_json(r.stdout)
    assert line["config"]["exchange"] == "rccl" and line["config"]["exchange_note"] is None


def test_a_failing_rccl_child_is_followed_by_a_host_exchange_child():
    r = run(["--gpus", "2", "--steps", "3"], "rccl_fails")
    assert r.returncode == 0, r.stderr
    line = last_json(r.stdout)
    assert line["config"]["exchange"] == "host"
    assert "rccl-exchange run exited with status 3" in line["config"]["exchange_note"], line
    assert "host exchange" in r.stderr


def test_a_hanging_rccl_child_is_killed_and_replaced():
    r = run(["--gpus", "2", "--steps", "3", "--child-timeout", "1.5"], "rccl_hangs")
    assert r.returncode == 0, r.stderr
    line = last_json(r.stdout)
    assert line["config"]["exchange"] == "host" and "was killed after" in line["config"]["exchange_note"], line


def test_both_runs_failing_is_a_failure_with_no_line():
    r = run(["--gpus", "2", "--steps", "3"], "always_fails")
    assert r.returncode == 1 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_an_explicit_host_exchange_has_no_second_attempt():
    r = run(["--gpus", "2", "--exchange", "host"], "always_fails")
    assert r.returncode == 1
    r = run(["--gpus", "2", "--exchange", "host"], "ok")
    assert last_json(r.stdout)["config"]["exchange"] == "host"


def test_one_gpu_runs_unsupervised_unless_asked():
    # (N = 1 without --supervise goes straight into the measurement, which needs a GPU: only the supervised form is tested here)
    r = run(["--gpus", "1", "--supervise", "--exchange", "rccl"], "rccl_fails")
    assert r.returncode == 0 and last_json(r.stdout)["config"]["exchange"] == "host"


def test_ranks_under_a_launcher_take_the_second_step_together():
    """Two supervisors as torch.distributed.run would start them (RANK / WORLD_SIZE / MASTER_PORT): rank 1's
    RCCL child fails, rank 0's succeeds -- both must go on to the host exchange (on the next port), and
    only rank 0 prints."""
    env = dict(os.environ, VT_BENCH_CHILD=FAKE, FAKE_CHILD="rccl_fails", FAKE_FAIL_RANKS="1", WORLD_SIZE="2", MASTER_PORT="29641",
               MASTER_ADDR="127.0.0.1", TORCHELASTIC_USE_AGENT_STORE="True")   # (what torch.distributed.run's agent puts there)
    procs = []
    for rank in (0, 1):
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "3"], env=dict(env, RANK=str(rank), LOCAL_RANK=str(rank)),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    line = last_json(outs[0][0])
    assert line["config"]["exchange"] == "host" and line["config"]["master_port"] == "29642", line
    # ... where the second run's rank 0 hosts the store itself: the launcher's agent only listens on the first port
    assert line["config"]["agent_store"] == "False", line
    assert "failed on another rank" in line["config"]["exchange_note"]
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]


def _driver_command(nproc, port, extra=()):
    """The driver's own launch line (the task's contract), at the width of the node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
            "--master-port", str(port), BENCH, "--gpus", str(nproc), "--steps", "20", "--warmup", "5"] + list(extra)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_eight_ranks_as_the_driver_launches_them_meet_and_print_one_line():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8 --steps 20 --warmup 5` on a CPU
    box: eight supervisors, eight children that really meet (the stand-in initialises a process group over the environment
    the supervisor hands down -- gloo here, RCCL on the node -- and all_gathers the ranks), ONE line from rank 0 that
    carries the width: n_gpus 8, rccl_ranks 8, the scaling the command asked for, the driver's steps and warmup.
    (The eight-GPU wire itself has never been available to this build: DESIGN 6.)"""
    env = dict(os.environ, VT_BENCH_CHILD=FAKE, FAKE_CHILD="meet", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "FAKE_FAIL_RANKS"):
        env.pop(k, None)
    r = subprocess.run(_driver_command(8, _free_port()), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = last_json(r.stdout)
    assert (line["n_gpus"], line["gpus_argument"], line["rccl_ranks"], line["met"]) == (8, 8, 8, list(range(8))), line
    assert (line["scaling"], line["steps"], line["warmup"]) == ("strong", 20, 5), line
    assert line["config"]["exchange"] == "rccl" and line["config"]["exchange_note"] is None, line


def test_eight_ranks_take_the_second_step_together_when_one_of_them_fails():
    """The same launch with rank 5's first child dying before the collective: its seven peers wait in the rendezvous until
    their timeout, every supervisor learns that the attempt failed somewhere, and all eight go on to the host exchange --
    on the next port, with a store of their own -- where they meet and rank 0 prints the line with the reason in it."""
    env = dict(os.environ, VT_BENCH_CHILD=FAKE, FAKE_CHILD="meet", FAKE_FAIL_RANKS="5", FAKE_MEET_TIMEOUT="20", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    port = _free_port()
    r = subprocess.run(_driver_command(8, port, ["--scaling", "weak"]), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = last_json(r.stdout)
    assert (line["n_gpus"], line["rccl_ranks"], line["met"], line["scaling"]) == (8, 8, list(range(8)), "weak"), line
    assert line["config"]["exchange"] == "host" and line["config"]["master_port"] == str(port + 1), line
    assert line["config"]["agent_store"] == "False" and "the rccl-exchange run" in line["config"]["exchange_note"], line


def test_json_line_of_takes_the_last_object():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.json_line_of('x\n{"a": 1}\nNCCL banner\n{"b": 2}\ntrailing') == {"b": 2}
    assert bench.json_line_of("nothing here") is None
