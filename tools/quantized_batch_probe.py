#!/usr/bin/env python3
"""tools/quantized_batch_probe.py -- quantized_search one by one against groups of up to eight per
sweep of the sign bits (vt_flat_quantized_search_batch), N = 10 M x 768 cosine by default.
One JSON line per batch size: ms per call, queries/s, the grouped Hamming pass's own time."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--candidates", type=int, default=100)
    a = ap.parse_args()
    import torch
    from vettore_amd import nifs
    dev = torch.device("cuda:0")
    x = bench.build_shard(torch, dev, a.rows, a.dim, bench.SEED_CORPUS)
    ref = nifs.flat_new_cosine()
    assert nifs.flat_load_device_matrix(ref, bench.doc_ids(0, a.rows), x.data_ptr(), a.rows, a.dim) == ("ok", ())
    del x
    qs = bench.normalized_queries(512, a.dim, bench.SEED_QUERY)
    nifs.flat_quantized_search(ref, qs[0], a.candidates, 10)
    for nq in (1, 2, 4, 8, 16, 64):
        reps = max(8, 256 // nq)
        # (one untimed call per batch size: the first call of a shape pays its buffers -- 13 ms once)
        if nq > 1:
            assert nifs.flat_quantized_search_batch(ref, qs[:nq], a.candidates, 10)[0] == "ok"
        nifs.flat_set_profiling(ref, True)
        nifs.flat_get_profile(ref, reset=True)
        t0 = time.perf_counter()
        for r in range(reps):
            q = qs[(r * nq) % 448:(r * nq) % 448 + nq]
            if nq == 1:
                st, _ = nifs.flat_quantized_search(ref, q[0], a.candidates, 10)
            else:
                st, _ = nifs.flat_quantized_search_batch(ref, q, a.candidates, 10)
            assert st == "ok"
        dt = time.perf_counter() - t0
        p = nifs.flat_get_profile(ref, reset=True)
        nifs.flat_set_profiling(ref, False)
        print(json.dumps({"batch": nq, "ms_per_call": dt / reps * 1e3, "queries_per_s": nq * reps / dt,
                          "hamming_launches": p["hamming_launches"], "hamming_ms_per_launch": p["hamming_ms"] / max(1, p["hamming_launches"]),
                          "GBps": p["hamming_bytes"] / max(1e-9, p["hamming_ms"]) / 1e6}), flush=True)


if __name__ == "__main__":
    main()
