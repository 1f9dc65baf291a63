#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: queries/sec (+ achieved HBM GB/s) of flat
cosine top-10 search, N=10M, d=768, single query in flight, on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--rows N] [--dim D]

A step = one `flat_search` call (one query scanned against the whole corpus)
through the C ABI of libvettore_hip.so: query H2D, scan + fused top-k kernel,
merge kernel, result D2H.  The corpus is resident in HBM before timing starts.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the SAME
10M-row corpus is row-sharded across the ranks (strong scaling); every query
runs on every shard and the per-shard top-k lists are merged after one
all_gather of fixed-size records (the path's only exchange step).

Synthetic data: iid uniform(-1,1) coordinates, rows
L2-normalised, 1% verbatim duplicate rows, ids "doc-<i>", seeds 20260721/22.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SEED_CORPUS, SEED_QUERY = 20260721, 20260722


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--limit", type=int, default=10)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--mode", choices=["single", "batch", "quantized", "funnel"], default="single",
                    help="single: BASELINE.json metric (default); batch: configs[2] (dot, 256-query batches, "
                         "FP32 MFMA); quantized: configs[4] (sign-bit Hamming pass + exact rerank)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the all_gather exchange even on one rank (prices the multi-GPU merge step)")
    ap.add_argument("--stages", default="128", help="funnel mode: prefix lengths (collection.ex:660-672 default min(d,128))")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--candidates", type=int, default=100)
    return ap.parse_args()


def doc_ids(start, count):
    """ids "doc-<i>" for i in [start+1, start+count] as (bytes, offsets)."""
    idx = np.arange(start + 1, start + count + 1, dtype=np.int64)
    digits = np.floor(np.log10(idx)).astype(np.int64) + 1
    off = np.zeros(count + 1, dtype=np.uintp)
    off[1:] = np.cumsum(digits + 4)
    blob = b"".join([b"doc-%d" % i for i in idx.tolist()])
    assert len(blob) == int(off[-1])
    return blob, off


def build_shard(torch, device, rows, dim, seed, chunk=1 << 20):
    """uniform(-1,1) rows, L2-normalised, 1% verbatim duplicates, generated in HBM."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.empty((rows, dim), dtype=torch.float32, device=device)
    for s in range(0, rows, chunk):
        e = min(rows, s + chunk)
        blk = torch.rand((e - s, dim), generator=g, device=device, dtype=torch.float32) * 2.0 - 1.0
        blk /= torch.linalg.vector_norm(blk.double(), dim=1, keepdim=True).float()
        x[s:e] = blk
        del blk
    ndup = rows // 100
    if ndup:
        src = torch.randint(0, rows, (ndup,), generator=g, device=device)
        dst = torch.randint(0, rows, (ndup,), generator=g, device=device)
        x[dst] = x[src]
    torch.cuda.synchronize()
    return x


def cpu_baseline(dim, limit, budget_s):
    """The oracle in the reference's own shape (hash map of separately allocated
    rows, per-row id clone, bounded heap: flat.rs:96-124) on a bounded sample of
    the same workload.  A reference search is single-threaded; its throughput
    comes from T callers searching concurrently under the read lock
    (nifs.rs:304-308), so both are timed: one thread, then T = host cores."""
    import threading
    import oracle
    rows = 200_000
    rng = np.random.default_rng(SEED_CORPUS)
    x = rng.uniform(-1.0, 1.0, size=(rows, dim)).astype(np.float32)
    x /= np.sqrt(np.sum(x.astype(np.float64) ** 2, axis=1, keepdims=True)).astype(np.float32)
    ids = [b"doc-%d" % (i + 1) for i in range(rows)]
    ix = oracle.FlatIndex(oracle.METRIC_CODE["cosine"])
    ix.insert_matrix(ids, x)
    qrng = np.random.default_rng(SEED_QUERY)
    qs = qrng.uniform(-1, 1, size=(64, dim)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    ix.search(qs[0], limit)
    # one thread
    t0 = time.perf_counter()
    done = 0
    while done < len(qs) and time.perf_counter() - t0 < budget_s * 0.5:
        ix.search(qs[done], limit)
        done += 1
    dt1 = time.perf_counter() - t0
    single = rows * done / dt1
    # T concurrent readers (ctypes releases the GIL inside the C search)
    threads = max(1, os.cpu_count() or 1)
    counts = [0] * threads
    stop = time.perf_counter() + budget_s * 0.5

    def reader(t):
        i = t
        while time.perf_counter() < stop:
            ix.search(qs[i % len(qs)], limit)
            counts[t] += 1
            i += 1

    t0 = time.perf_counter()
    ths = [threading.Thread(target=reader, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dtn = time.perf_counter() - t0
    multi = rows * sum(counts) / dtn
    return {"rows_per_s_1": single, "rows_per_s_T": multi, "threads": threads, "sample_rows": rows,
            "queries_1": done, "queries_T": sum(counts), "seconds": dt1 + dtn}


def measured_read_peak():
    """GB/s of the plain read-only streaming kernel on this pool's MI355X
    (tools/hbm_peak.hip, recorded in profiles/r01_hbm_peak.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_peak.json")) as f:
            return float(json.load(f)["read_peak_GBps"])
    except (OSError, ValueError, KeyError):
        return None


def pmc_traffic(rows, dim):
    """HBM bytes per scan launch from the committed rocprofv3 PMC passes
    (profiles/pmc_latest.json: FETCH_SIZE and WRITE_SIZE collected in separate
    --pmc runs of this script, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950).  bench.py cannot read counters itself, so the figure
    is reported only when the profile was taken on the same workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            p = json.load(f)
        if p.get("rows") == rows and p.get("dim") == dim:
            return p["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def run_side_mode(a, torch, nifs, device):
    """configs[2] / configs[4] on one GPU: same corpus generator, own JSON line."""
    import ctypes as C
    from vettore_amd import _lib
    L = _lib.load()
    batch = a.mode == "batch"
    x = build_shard(torch, device, a.rows, a.dim, SEED_CORPUS)
    if batch:  # configs[2]: metric :dot -> :inner_product, no normalisation (collection.ex:1302, :1319)
        x.mul_(torch.empty((a.rows, 1), device=device).uniform_(8.0, 24.0))
        ref = nifs.flat_new_inner_product()
    else:
        ref = nifs.flat_new_cosine()
    assert nifs.flat_load_device_matrix(ref, doc_ids(0, a.rows), x.data_ptr(), a.rows, a.dim) == ("ok", ())
    del x
    torch.cuda.empty_cache()
    qrng = np.random.default_rng(SEED_QUERY)
    per = a.batch if batch else 1
    nq = (a.steps + a.warmup) * per
    qs = qrng.uniform(-1, 1, size=(nq, a.dim)).astype(np.float32)
    if not batch:
        qs /= np.linalg.norm(qs.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    outs = (C.c_void_p * per)()
    stage_list = [int(v) for v in a.stages.split(",")]
    stages = (C.c_size_t * len(stage_list))(*stage_list)

    def step(i):
        q = qs[i * per:(i + 1) * per]
        qp = q.ctypes.data_as(C.POINTER(C.c_float))
        if batch:
            assert L.vt_flat_search_batch(ref.handle, qp, per, a.dim, a.limit, outs) == 0
            for j in range(per):
                L.vt_hits_free(C.c_void_p(outs[j]))
        elif a.mode == "funnel":
            h = C.c_void_p()
            assert L.vt_flat_funnel_search(ref.handle, qp, a.dim, stages, len(stage_list), a.candidates, a.limit,
                                           C.byref(h)) == 0
            L.vt_hits_free(h)
        else:
            h = C.c_void_p()
            assert L.vt_flat_quantized_search(ref.handle, qp, a.dim, a.candidates, a.limit, C.byref(h)) == 0
            L.vt_hits_free(h)

    for i in range(a.warmup):
        step(i)
    nifs.flat_set_profiling(ref, True)
    nifs.flat_get_profile(ref, reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + a.steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = nifs.flat_get_profile(ref, reset=True)
    out = {
        "value": a.steps * per / dt, "unit": "queries/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if batch else "u64", "data": "synthetic",
    }
    if batch:
        ms = p["batch_ms"] / max(1, p["batch_launches"])
        tf = p["batch_flops"] / max(1e-9, p["batch_ms"]) / 1e9
        out["metric"] = "queries/sec, flat dot top-%d, N=%d d=%d, batch=%d" % (a.limit, a.rows, a.dim, per)
        out["config"] = {"workload": "index: :flat, metric: :dot, d=%d, N=%d, batch=%d queries (MFMA Q x D^T + exact rescoring)"
                         % (a.dim, a.rows, per), "fallback_queries": p["batch_fallbacks"]}
        out["roofline"] = {"bound": "mfma", "kernel": "mfma_scores_kernel", "achieved": tf, "peak": 157.3,
                           "unit": "TFLOP/s", "frac": tf / 157.3, "traffic": None, "avg_launch_ms": ms,
                           "algorithmic_flops_per_launch": p["batch_flops"] / max(1, p["batch_launches"])}
    elif a.mode == "funnel":
        ms = p["prefix_ms"] / max(1, p["prefix_launches"])
        gbs = p["prefix_bytes"] / max(1, p["prefix_launches"]) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out["dtype"] = "f64"
        out["metric"] = "queries/sec, funnel_search (f64 cosine on prefix %s, keep %d, exact rerank top-%d), N=%d d=%d" % (
            a.stages, a.candidates, a.limit, a.rows, a.dim)
        out["config"] = {"workload": "funnel_search stages=[%s] candidates=%d limit=%d, d=%d, N=%d" % (
            a.stages, a.candidates, a.limit, a.dim, a.rows)}
        out["roofline"] = {"bound": "hbm", "kernel": "cosine_scan_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ms,
                           "algorithmic_bytes_per_launch": p["prefix_bytes"] / max(1, p["prefix_launches"]),
                           "note": "useful bytes = rows * prefix * 4; the prefix of a 3 KiB row is a strided read"}
    else:
        ms = p["hamming_ms"] / max(1, p["hamming_launches"])
        gbs = p["hamming_bytes"] / max(1, p["hamming_launches"]) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out["metric"] = "queries/sec, quantized_search (sign-bit Hamming top-%d + exact cosine rerank top-%d), N=%d d=%d" % (
            a.candidates, a.limit, a.rows, a.dim)
        out["config"] = {"workload": "quantized_search candidates=%d limit=%d, d=%d, N=%d" % (a.candidates, a.limit, a.dim, a.rows)}
        out["roofline"] = {"bound": "hbm", "kernel": "hamming_dist_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ms,
                           "algorithmic_bytes_per_launch": p["hamming_bytes"] / max(1, p["hamming_launches"])}
    print(json.dumps(out))


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % a.gpus)
        a.gpus = world

    import torch  # first: its bundled libamdhip64 must be the one the process shares
    import torch.distributed as dist
    from vettore_amd import nifs, _lib
    from vettore_amd.sharded import ShardedFlat

    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_exchange
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    nifs.set_device(local_rank)
    if a.mode != "single":
        if world > 1:
            sys.exit("--mode %s is a single-GPU measurement" % a.mode)
        if a.mode == "batch" and a.steps == 200:
            a.steps, a.warmup = 8, 2
        return run_side_mode(a, torch, nifs, device)

    # ---- corpus: this rank's row block of the N-row corpus ------------------
    per = a.rows // world
    start = rank * per
    count = per if rank < world - 1 else a.rows - start
    t_build = time.perf_counter()
    x = build_shard(torch, device, count, a.dim, SEED_CORPUS + rank)
    ids = doc_ids(start, count)
    ref = nifs.flat_new_cosine()
    res = nifs.flat_load_device_matrix(ref, ids, x.data_ptr(), count, a.dim)
    assert res == ("ok", ()), res
    del x
    torch.cuda.empty_cache()
    sharded = ShardedFlat(ref, dist if use_dist else None, device, force_exchange=a.force_exchange)
    if use_dist and os.environ.get("VT_HOST_EXCHANGE") is None:
        # one ordering of all ids -> shard keys compare on the device (see vettore_amd/sharded.py)
        sharded.enable_device_exchange(ids, max_limit=max(a.limit, 16))

    qrng = np.random.default_rng(SEED_QUERY)
    nq = a.steps + a.warmup
    qs = qrng.uniform(-1, 1, size=(nq, a.dim)).astype(np.float32)
    qs /= np.linalg.norm(qs.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    # first search also sorts the ids (id-rank column) -- setup, not a step
    sharded.search(qs[0], a.limit)
    t_build = time.perf_counter() - t_build

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:
        search = lambda q: sharded.search(q, a.limit)  # noqa: E731
    else:
        # one rank, no exchange: the step is the C-ABI call itself (the hit list is
        # freed, not unpacked into Python objects -- a NIF would build BEAM terms here)
        L = _lib.load()
        hp = C.c_void_p()

        def search(q):
            st = L.vt_flat_search(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), a.dim, a.limit, C.byref(hp))
            assert st == 0, st
            n_hits = L.vt_hits_len(hp)
            L.vt_hits_free(hp)
            return range(n_hits)

    for i in range(a.warmup):
        search(qs[i])
    nifs.flat_set_profiling(ref, True)
    nifs.flat_get_profile(ref, reset=True)
    sync()
    t0 = time.perf_counter()
    for i in range(a.warmup, nq):
        hits = search(qs[i])
    sync()
    dt = time.perf_counter() - t0
    prof = nifs.flat_get_profile(ref, reset=True)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert len(hits) == min(a.limit, a.rows)

    if rank == 0:
        qps = a.steps / dt
        scan_ms = prof["scan_ms"] / max(1, prof["scan_launches"])
        bytes_per_launch = prof["scan_bytes"] / max(1, prof["scan_launches"])
        achieved = bytes_per_launch / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        out = {
            "metric": "queries/sec, flat cosine top-%d, N=%d d=%d (achieved HBM GB/s in roofline)" % (a.limit, a.rows, a.dim),
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "index: :flat, metric: :cosine, d=%d, N=%d, limit=%d, single query in flight" % (a.dim, a.rows, a.limit),
                "rows_per_gpu": count,
                "reduce_order": "pair",
                "sharding": ("row blocks, all_gather of per-shard top-k (%s merge)" % ("device" if sharded._dev else "host"))
                if use_dist else "none",
                "setup_s": round(t_build, 1),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "scan_topk_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(count, a.dim),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": scan_ms,
                "measured_read_peak": measured_read_peak(),
                "frac_of_measured_read_peak": (achieved / measured_read_peak()) if measured_read_peak() else None,
            },
        }
        if world == 1 and not a.no_cpu and a.cpu_seconds > 0:
            cb = cpu_baseline(a.dim, a.limit, a.cpu_seconds)
            out["cpu_baseline"] = {
                "value": cb["rows_per_s_T"] / a.rows,
                "unit": "queries/s",
                "cores": cb["threads"],
                "kind": "port",
                "sample": "%d concurrent readers, %d queries over %d rows x %d in %.1f s (plus %d single-thread queries); "
                          "rows/s scaled to N=%d" % (cb["threads"], cb["queries_T"], cb["sample_rows"], a.dim,
                                                     cb["seconds"], cb["queries_1"], a.rows),
                "single_thread_value": cb["rows_per_s_1"] / a.rows,
                "effective_GBps": cb["rows_per_s_T"] * a.dim * 4 / 1e9,
                "single_thread_effective_GBps": cb["rows_per_s_1"] * a.dim * 4 / 1e9,
            }
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL's version banner sits in libc's stdout buffer: flush it first so
        # that the JSON line is the last line of output
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
