#!/usr/bin/env python3
"""tools/funnel_batch_probe.py -- funnel_search one by one against groups of up to eight per sweep of
the prefixes (vt_flat_funnel_search_batch), N = 10 M x 768 cosine, stages [128], candidates 100.
One JSON line per batch size: ms per call, queries/s, the grouped prefix pass's own time."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--prefix", type=int, default=128)
    ap.add_argument("--candidates", type=int, default=100)
    a = ap.parse_args()
    import torch
    from vettore_amd import nifs
    x = bench.build_shard(torch, torch.device("cuda:0"), a.rows, a.dim, bench.SEED_CORPUS)
    ref = nifs.flat_new_cosine()
    assert nifs.flat_load_device_matrix(ref, bench.doc_ids(0, a.rows), x.data_ptr(), a.rows, a.dim) == ("ok", ())
    del x
    qs = bench.normalized_queries(512, a.dim, bench.SEED_QUERY)
    stages = [a.prefix]
    for nq in (1, 2, 4, 8, 16, 64):
        reps = max(8, 128 // nq)
        call = (lambda q: nifs.flat_funnel_search(ref, q[0], stages, a.candidates, 10)) if nq == 1 else \
               (lambda q: nifs.flat_funnel_search_batch(ref, q, stages, a.candidates, 10))
        assert call(qs[:nq])[0] == "ok"   # (the first call of a shape pays its buffers)
        nifs.flat_set_profiling(ref, True)
        nifs.flat_get_profile(ref, reset=True)
        t0 = time.perf_counter()
        for r in range(reps):
            assert call(qs[(r * nq) % 448:(r * nq) % 448 + nq])[0] == "ok"
        dt = time.perf_counter() - t0
        p = nifs.flat_get_profile(ref, reset=True)
        nifs.flat_set_profiling(ref, False)
        print(json.dumps({"batch": nq, "ms_per_call": round(dt / reps * 1e3, 4), "queries_per_s": round(nq * reps / dt, 1),
                          "prefix_launches": p["prefix_launches"], "prefix_queries": p["prefix_queries"],
                          "prefix_ms_per_launch": round(p["prefix_ms"] / max(1, p["prefix_launches"]), 4),
                          "prefix_GBps": round(p["prefix_bytes"] / max(1e-9, p["prefix_ms"]) / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
