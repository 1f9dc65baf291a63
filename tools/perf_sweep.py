#!/usr/bin/env python3
"""Kernel-level sweep on one MI355X: scan kernel GB/s for several shapes,
limits and metrics, hamming scan GB/s, end-to-end latency.  Diagnostic only."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: shares its HIP runtime)
from vettore_amd import nifs  # noqa: E402
from bench import build_shard, doc_ids  # noqa: E402


def run(metric, rows, dim, limit, quantized=False, steps=30, order=0):
    dev = torch.device("cuda", 0)
    x = build_shard(torch, dev, rows, dim, 1234)
    ref = nifs._flat_new(metric)
    nifs.flat_set_reduce_order(ref, order)
    assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
    del x
    torch.cuda.empty_cache()
    rng = np.random.default_rng(1)
    qs = rng.uniform(-1, 1, size=(steps + 3, dim)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    call = (lambda q: nifs.flat_quantized_search(ref, q, 100, limit)) if quantized else (lambda q: nifs.flat_search(ref, q, limit))
    for i in range(3):
        assert call(qs[i])[0] == "ok"
    nifs.flat_set_profiling(ref, True)
    nifs.flat_get_profile(ref, reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3, steps + 3):
        call(qs[i])
    dt = (time.perf_counter() - t0) / steps
    p = nifs.flat_get_profile(ref, reset=True)
    out = {"metric": nifs.METRICS[metric], "rows": rows, "dim": dim, "limit": limit, "quantized": quantized,
           "order": order, "e2e_ms": round(dt * 1e3, 4)}
    if p["scan_launches"]:
        ms = p["scan_ms"] / p["scan_launches"]
        out["scan_ms"] = round(ms, 4)
        out["scan_GBps"] = round(p["scan_bytes"] / p["scan_launches"] / ms / 1e6, 1)
        out["scan_launches_per_query"] = p["scan_launches"] / steps
    if p["hamming_launches"]:
        ms = p["hamming_ms"] / p["hamming_launches"]
        out["hamming_ms"] = round(ms, 4)
        out["hamming_GBps"] = round(p["hamming_bytes"] / p["hamming_launches"] / ms / 1e6, 1)
    print(json.dumps(out), flush=True)
    del ref


def run_batch(metric, rows, dim, nq, limit, normalize, steps=4):
    dev = torch.device("cuda", 0)
    x = build_shard(torch, dev, rows, dim, 1234)
    if not normalize:
        x.mul_(torch.linalg.vector_norm(torch.rand((rows, 1), device=dev) + 0.5, dim=1, keepdim=True) * 16)
    ref = nifs._flat_new(metric)
    assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
    del x
    torch.cuda.empty_cache()
    rng = np.random.default_rng(1)
    qs = rng.uniform(-1, 1, size=(nq, dim)).astype(np.float32)
    if normalize:
        qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    import ctypes as C
    from vettore_amd import _lib
    L = _lib.load()
    outs = (C.c_void_p * nq)()
    qp = qs.ctypes.data_as(C.POINTER(C.c_float))

    def call():  # the C ABI call itself; hit lists are freed, not unpacked into Python objects
        assert L.vt_flat_search_batch(ref.handle, qp, nq, dim, limit, outs) == 0
        for i in range(nq):
            L.vt_hits_free(C.c_void_p(outs[i]))

    call()
    nifs.flat_set_profiling(ref, True)
    nifs.flat_get_profile(ref, reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        call()
    dt = (time.perf_counter() - t0) / steps
    p = nifs.flat_get_profile(ref, reset=True)
    out = {"batch": nq, "metric": nifs.METRICS[metric], "rows": rows, "dim": dim, "limit": limit,
           "e2e_ms_per_batch": round(dt * 1e3, 3), "qps": round(nq / dt, 1),
           "mfma_ms": round(p["batch_ms"] / max(1, p["batch_launches"]), 3),
           "mfma_TFLOPs": round(p["batch_flops"] / max(1e-9, p["batch_ms"]) / 1e9, 1),
           "fallbacks": p["batch_fallbacks"], "queries": p["batch_queries"]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    R = 2_000_000
    if which in ("all", "shapes"):
        for dim in (768, 384, 1024, 1536, 128, 100):
            run(2, R * 768 // dim, dim, 10)
    if which in ("all", "metrics"):
        for m in (0, 1, 3, 4, 5, 6, 7, 8):
            run(m, R, 768, 10)
        for o in (1, 2):
            run(2, R, 768, 10, order=o)
    if which in ("all", "limits"):
        for k in (1, 64, 100, 256, 300, 1000):
            run(2, R, 768, k)
    if which in ("all", "quantized"):
        run(2, R, 768, 10, quantized=True)
        run(2, 10_000_000, 768, 10, quantized=True)
    if which in ("all", "batch"):
        run_batch(3, 2_000_000, 768, 256, 10, False)
        run_batch(2, 2_000_000, 768, 256, 10, True)
        run_batch(3, 2_000_000, 768, 64, 10, False)
        run_batch(3, 10_000_000, 768, 256, 10, False, steps=3)
    if which in ("all", "configs"):   # BASELINE.json configs 1, 2, 4 (per-GPU shard) on one GPU
        run(2, 10_000, 384, 10, steps=300)
        run(2, 1_000_000, 768, 10, steps=100)
        run(0, 5_000_000, 768, 10, steps=40)
        run_batch(0, 5_000_000, 768, 256, 10, False, steps=3)
        run_batch(2, 10_000_000, 768, 256, 10, True, steps=3)
    if which in ("all", "small"):
        for rows in (10_000, 100_000, 1_000_000):
            run(2, rows, 768, 10, steps=100)
