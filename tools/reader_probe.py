#!/usr/bin/env python3
"""Throughput of T threads searching ONE handle, with and without the coalescing of searches that
meet on it (`VT_COALESCE=0`): queries/s, mean latency, and how many searches travelled in batches.
    ROWS=10000000 DIM=768 THREADS=1,2,4,8,16,32,64 SECONDS=3 python tools/reader_probe.py"""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vettore_amd import nifs, _lib  # noqa: E402
from bench import build_shard, doc_ids  # noqa: E402

L = _lib.load()


def run(ref, qs, threads, seconds, d, limit):
    stop = threading.Event()
    counts = [0] * threads
    lat = [0.0] * threads

    def worker(t):
        h = C.c_void_p()
        i = t
        while not stop.is_set():
            q = qs[i % len(qs)]
            t0 = time.perf_counter()
            st = L.vt_flat_search(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), d, limit, C.byref(h))
            lat[t] += time.perf_counter() - t0
            assert st == 0, st
            L.vt_hits_free(h)
            counts[t] += 1
            i += threads

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    b0 = nifs.flat_coalesce_stats(ref)
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    time.sleep(seconds)
    stop.set()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    b1 = nifs.flat_coalesce_stats(ref)
    total = sum(counts)
    return {"qps": round(total / dt, 1), "mean_latency_ms": round(sum(lat) / max(1, total) * 1e3, 3),
            "batches": b1[0] - b0[0], "in_batches": b1[1] - b0[1], "searches": total}


def main():
    n, d = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("DIM", 768))
    seconds = float(os.environ.get("SECONDS", 3))
    limit = int(os.environ.get("LIMIT", 10))
    metric = int(os.environ.get("METRIC", 2))
    shards = int(os.environ.get("SHARDS", 0))
    dev = torch.device("cuda", 0)
    x = build_shard(torch, dev, n, d, 5)
    ref = nifs.flat_new_sharded(metric, [0] * shards) if shards else nifs._flat_new(metric)
    assert nifs.flat_load_device_matrix(ref, doc_ids(0, n), x.data_ptr(), n, d) == ("ok", ())
    del x
    torch.cuda.empty_cache()
    rng = np.random.default_rng(3)
    qs = rng.uniform(-1, 1, (512, d)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    qs = [np.ascontiguousarray(q) for q in qs]
    nifs.flat_search(ref, qs[0], limit)
    for threads in (int(v) for v in os.environ.get("THREADS", "1,2,4,8,16,32,64").split(",")):
        row = {"rows": n, "dim": d, "threads": threads}
        for label, env in (("coalesced", None), ("side_by_side", "0")):
            nifs.debug_set("coalesce", 1 if env is None else int(env))
            row[label] = run(ref, qs, threads, seconds, d, limit)
        nifs.debug_set("coalesce", 1)
        row["gain"] = round(row["coalesced"]["qps"] / max(1e-9, row["side_by_side"]["qps"]), 2)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
