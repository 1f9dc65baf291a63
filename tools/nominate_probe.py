#!/usr/bin/env python3
"""tools/nominate_probe.py -- K2 (FP32 matrix cores) against K2b (bf16 nomination) on one corpus.

  python tools/nominate_probe.py [--rows N] [--dim D] [--batch B] [--batches K] [--metric dot|cosine|l2]

For each mode: K batches of B queries through vt_flat_search_batch, the dominant kernel's own
time from the library's HIP events, fallbacks / second passes / candidates per query, and every
query of the first batch compared bit for bit with its own vt_flat_search.  One JSON line per mode.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (corpus generator, ids)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--limit", type=int, default=10)
    ap.add_argument("--metric", default="dot", choices=["dot", "cosine", "l2"])
    ap.add_argument("--modes", default="f32,bf16")
    ap.add_argument("--verify", type=int, default=256, help="queries of the first batch checked against single searches")
    a = ap.parse_args()
    import torch
    from vettore_amd import _lib, nifs
    L = _lib.load()
    dev = torch.device("cuda:0")
    x = bench.build_shard(torch, dev, a.rows, a.dim, bench.SEED_CORPUS)
    if a.metric == "dot":  # config 3: no normalisation (collection.ex:1302, :1319)
        x.mul_(torch.empty((a.rows, 1), device=dev).uniform_(8.0, 24.0))
        ref = nifs.flat_new_inner_product()
    elif a.metric == "l2":
        x.mul_(torch.empty((a.rows, 1), device=dev).uniform_(0.5, 2.0))
        ref = nifs.flat_new_l2()
    else:
        ref = nifs.flat_new_cosine()
    assert nifs.flat_load_device_matrix(ref, bench.doc_ids(0, a.rows), x.data_ptr(), a.rows, a.dim) == ("ok", ())
    del x
    torch.cuda.empty_cache()
    nq = a.batch * (a.batches + 1)
    rng = np.random.default_rng(bench.SEED_QUERY)
    qs = rng.uniform(-1, 1, size=(nq, a.dim)).astype(np.float32)
    if a.metric == "cosine":
        qs /= np.linalg.norm(qs.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    outs = (C.c_void_p * a.batch)()

    def batch(i, keep=False):
        q = qs[i * a.batch:(i + 1) * a.batch]
        assert L.vt_flat_search_batch(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), a.batch, a.dim, a.limit, outs) == 0
        res = [C.c_void_p(outs[j]) for j in range(a.batch)]
        if keep:
            return [bench.hits_of(L, r) for r in res]
        for r in res:
            L.vt_hits_free(r)

    singles = None
    for mode in a.modes.split(","):
        assert nifs.flat_set_batch_nominate(ref, {"f32": _lib.NOMINATE_F32, "bf16": _lib.NOMINATE_BF16}[mode]) == "ok"
        first = batch(0, keep=True)  # warm-up, kept for the comparison
        nifs.flat_set_profiling(ref, True)
        nifs.flat_get_profile(ref, reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1, a.batches + 1):
            batch(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        p = nifs.flat_get_profile(ref, reset=True)
        nifs.flat_set_profiling(ref, False)
        if singles is None:
            singles = []
            for j in range(min(a.verify, a.batch)):
                h = C.c_void_p()
                assert L.vt_flat_search(ref.handle, qs[j].ctypes.data_as(C.POINTER(C.c_float)), a.dim, a.limit, C.byref(h)) == 0
                singles.append(bench.hits_of(L, h))
        wrong = sum(1 for j, s in enumerate(singles) if first[j] != s)
        key = "nominate" if mode.startswith("bf16") else "batch"
        launches = max(1, p[key + "_launches"])
        ms = p[key + "_ms"] / launches
        out = {"mode": mode, "metric": a.metric, "rows": a.rows, "dim": a.dim, "batch": a.batch, "batches": a.batches,
               "ms_per_batch": dt / a.batches * 1e3, "queries_per_s": a.batch * a.batches / dt,
               "kernel_ms": ms, "launches": p[key + "_launches"], "fallbacks": p["batch_fallbacks"],
               "wrong_of_verified": [wrong, len(singles)]}
        if mode.startswith("bf16"):
            out["GBps"] = p["nominate_bytes"] / launches / (ms * 1e-3) / 1e9 if ms else 0.0
            out["TFLOPs"] = p["nominate_flops"] / launches / (ms * 1e-3) / 1e12 if ms else 0.0
            out["second_passes"] = p["nominate_second_passes"]
            out["candidates_per_query"] = p["nominate_candidates"] / max(1, p["nominate_queries"])
        else:
            out["TFLOPs"] = p["batch_flops"] / launches / (ms * 1e-3) / 1e12 if ms else 0.0
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
