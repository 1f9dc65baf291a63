"""bench.py's multi-GPU forms on the one-GPU boxes of this pool: the supervisor's
fallback driven by a real wedged all-gather, weak scaling, the batched leg on a sharded handle and
between ranks.  What no test here can exercise is the wire between two devices."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
HOOKS = os.path.join(ROOT, "vettore_amd", "lib", "libvettore_hip_hooks.so")


def bench(args, timeout=600, launcher=None, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher), "--master-addr", "127.0.0.1",
                "--master-port", "29713"]
    if "--release-wait" not in args:
        args = list(args) + ["--release-wait", "0"]      # (setup's pause is for measurements; these runs check behaviour)
    r = subprocess.run(cmd + [BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_a_wedged_all_gather_ends_in_a_host_exchange_line():
    """The RCCL child meets an all-gather that does not complete (the stall hook of libvettore_hip_hooks.so
    in front of it, a 300-ms deadline behind it): the library fails the search with its message, the
    child exits non-zero, the supervisor starts the host-exchange child and prints ITS line."""
    assert os.path.exists(HOOKS)
    r, line = bench(["--gpus", "1", "--supervise", "--exchange", "rccl", "--rows", "200000", "--steps", "20", "--warmup", "3",
                     "--no-cpu", "--no-side"],
                    VETTORE_HIP_LIB=HOOKS, VT_TEST_EXCHANGE_STALL_MS=2000, VT_EXCHANGE_TIMEOUT_MS=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "RCCL exchange timed out" in r.stderr
    note = line["config"]["exchange_note"]
    assert "rccl-exchange run exited with status" in note and "host exchange" in note, line
    assert line["value"] > 0 and line["n_gpus"] == 1


def test_a_healthy_one_rank_rccl_run_needs_no_second_child():
    r, line = bench(["--gpus", "1", "--supervise", "--exchange", "rccl", "--rows", "200000", "--steps", "20", "--warmup", "3", "--no-cpu",
                     "--no-side"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["rccl_ranks"] == 1 and "this line is the run over the host exchange" not in line["config"].get("exchange_note", "")
    assert "exchange_ms" in line["config"] and "per_shard_scan_ms" in line["config"]


def test_weak_scaling_and_the_batched_leg_on_a_two_shard_handle():
    r, line = bench(["--gpus", "2", "--devices", "0,0", "--scaling", "weak", "--rows", "150000", "--steps", "10", "--warmup", "2", "--no-cpu"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["scaling"] == "weak" and "N=300000" in line["metric"] and line["n_gpus"] == 2, line
    r, line = bench(["--gpus", "2", "--devices", "0,0", "--mode", "batch", "--metric", "l2", "--batch", "64", "--rows", "300000",
                     "--steps", "4", "--warmup", "1", "--no-cpu", "--debug-set", "force_batch_mfma=1"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["config"]["verified"] and "batch=64" in line["metric"] and line["roofline"]["kernel"] == "shadow_scores_kernel", line
    assert line["config"]["per_shard_pass_ms"] > 0


def test_two_ranks_under_the_launcher_over_the_host_exchange():
    """torch.distributed.run with two ranks on the one GPU: RCCL cannot serve that, so the line comes from the
    gloo / 64-byte-record exchange -- single queries and the batched leg."""
    r, line = bench(["--gpus", "2", "--devices", "0,0", "--exchange", "host", "--rows", "200000", "--steps", "10", "--warmup", "2", "--no-cpu"],
                    launcher=2)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["config"]["processes"] == 2 and "gloo" in line["config"]["sharding"], line
    r, line = bench(["--gpus", "2", "--devices", "0,0", "--exchange", "host", "--mode", "batch", "--metric", "l2", "--batch", "32",
                     "--rows", "200000", "--steps", "3", "--warmup", "1", "--no-cpu", "--debug-set", "force_batch_mfma=1"], launcher=2)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["config"]["verified"] and line["config"]["processes"] == 2, line


def test_the_fallback_under_the_launcher_meets_on_a_store_of_its_own():
    """The driver's own form -- torch.distributed.run, default exchange -- with both ranks on the one GPU: RCCL refuses
    two ranks on a device, every rank's first child exits non-zero, and the second children must find each other.
    Under the launcher the store at MASTER_PORT belongs to the launcher's agent (workers are told to come as clients);
    the second run shifts the port, so its rank 0 has to host the store itself -- when it did not, both ranks waited
    2 x 120 s for a listener that never came and the run ended without a line."""
    r, line = bench(["--gpus", "2", "--devices", "0,0", "--rows", "200000", "--steps", "10", "--warmup", "2", "--no-cpu"], launcher=2, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line is not None and line["n_gpus"] == 2 and line["config"]["processes"] == 2, line
    assert "rccl-exchange run exited with status" in line["config"]["exchange_note"], line
    assert "gloo" in line["config"]["sharding"], line


def test_the_width_of_the_node_rehearsed_on_one_gpu_carries_config_4():
    """`bench.py --gpus 8` also measures BASELINE configs[3] (L2, rows sharded over eight GPUs, single queries and 4 096 in
    one call) as side.config4 -- the driver passes no flags, so the one 8-GPU run there may ever be has to carry it.  Here:
    eight shards on the one card, small sizes; the leg's keys, its own roofline, both checks, the summary."""
    r, line = bench(["--gpus", "8", "--devices", "0,0,0,0,0,0,0,0", "--rows", "4000000", "--config4-rows", "250000", "--steps", "20",
                     "--warmup", "5", "--no-cpu"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 8 and line["config"]["devices"] == [0] * 8, line
    leg = line["side"]["config4"]
    assert leg["n_gpus"] == 8 and leg["rows_per_gpu"] == 250000 and "metric: :l2" in leg["workload"] and "N=2000000" in leg["workload"], leg
    assert leg["verified"] and leg["checks"]["batched_list_equals_single_search"] and leg["checks"]["brute_force_over_every_shard"], leg
    assert leg["single"]["steps"] == 200 and leg["single"]["value"] > 0 and leg["batch_4096_one_call"]["value"] > 0, leg
    roof = leg["single"]["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] <= 1 and roof["algorithmic_bytes_per_launch"] > 0, roof
    print("side.config4 rehearsed in %.1f s: %s" % (leg["seconds"], {k: leg[k] for k in ("single", "batch_4096_one_call")}))
    assert line["summary"]["config4_verified"] is True and line["summary"]["config4_batch_4096_queries_per_s"] == leg["batch_4096_one_call"]["value"]


def test_eight_ranks_under_the_launcher_on_one_gpu_skip_config_4_over_the_host_exchange():
    """The driver's own command for N = 8 -- torch.distributed.run, eight ranks -- on the one GPU of this box: RCCL cannot
    serve eight ranks on one device, so the line comes from the host-exchange children (gloo), eight real processes that
    each hold a shard on the card; that fallback says that it skipped the config 4 leg instead of measuring something else
    under its name."""
    r, line = bench(["--gpus", "8", "--devices", "0,0,0,0,0,0,0,0", "--exchange", "host", "--rows", "800000", "--config4-rows", "50000",
                     "--steps", "5", "--warmup", "2", "--no-cpu"], launcher=8, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 8 and line["config"]["processes"] == 8 and "gloo" in line["config"]["sharding"], line
    assert "skipped" in line["side"]["config4"], line["side"]


def test_config_4_between_rank_processes_rehearsed_with_two_ranks():
    """side.config4 in its other form -- one rank process per GPU, records exchanged between the ranks, what the driver's
    launch at N = 8 runs -- with two rank processes on the one GPU over gloo (`--config4-anyway`: the leg at this width and
    over this exchange is for this rehearsal only; RCCL itself wants a device per rank): both ranks take part in every
    collective of the leg, the batched lists equal the single searches, the brute-force pass over each rank's own rows finds
    nothing closer, rank 0 prints one line."""
    r, line = bench(["--gpus", "2", "--devices", "0,0", "--exchange", "host", "--rows", "300000", "--config4-rows", "150000", "--config4-anyway",
                     "--steps", "5", "--warmup", "2", "--no-cpu"], launcher=2, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    leg = line["side"]["config4"]
    assert "failed" not in leg and "skipped" not in leg, leg
    assert leg["n_gpus"] == 2 and leg["rows_per_gpu"] == 150000 and "N=300000" in leg["workload"] and "gloo" in leg["exchange"], leg
    assert leg["verified"] and leg["checks"]["batched_list_equals_single_search"] and leg["checks"]["brute_force_over_every_shard"], leg
    assert leg["single"]["roofline"]["algorithmic_bytes_per_launch"] == 150000 * 768 * 4, leg["single"]


def test_the_rank_per_gpu_path_over_rccl_with_one_rank():
    """What every rank of the driver's N > 1 launch runs -- torch.distributed over nccl (= RCCL), the global id ranking
    installed as the shard's rank column, per query vt_flat_search_begin into a device block, all_gather_into_tensor on
    the library's stream, vt_flat_merge_gathered -- with the one rank a one-GPU box can give it (`--force-exchange`): the
    collective really is RCCL's, only the peers are missing."""
    r, line = bench(["--gpus", "1", "--force-exchange", "--rows", "300000", "--steps", "20", "--warmup", "5", "--no-cpu", "--no-side"],
                    MASTER_ADDR="127.0.0.1", MASTER_PORT="29733")
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["value"] > 0, line
    assert "all_gather of per-shard top-k over RCCL (device merge)" in line["config"]["sharding"], line["config"]
    assert line["config"]["processes"] == 1 and 0 < line["roofline"]["frac"] <= 1, line
    # ... and its batched leg: every rank answers the whole batch on its rows, ONE all_gather of wire blocks over RCCL
    r, line = bench(["--gpus", "1", "--force-exchange", "--mode", "batch", "--metric", "l2", "--batch", "64", "--rows", "300000", "--steps", "3",
                     "--warmup", "1", "--no-cpu", "--debug-set", "force_batch_mfma=1"], MASTER_ADDR="127.0.0.1", MASTER_PORT="29734")
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["config"]["verified"] and line["rccl_ranks"] == 1 and "RCCL" in line["config"]["sharding"], line
    # ... and side.config4 in the form the driver's launch at N = 8 runs it, records gathered over RCCL
    r, line = bench(["--gpus", "1", "--force-exchange", "--config4-anyway", "--config4-rows", "200000", "--rows", "200000", "--steps", "5",
                     "--warmup", "2", "--no-cpu"], MASTER_ADDR="127.0.0.1", MASTER_PORT="29735")
    assert r.returncode == 0, r.stderr[-3000:]
    leg = line["side"]["config4"]
    assert "failed" not in leg and leg["verified"] and leg["rccl_ranks"] == 1 and "over RCCL" in leg["exchange"], leg
