#!/usr/bin/env python3
"""Cost of search / quantized / funnel / hybrid on a plain handle against handles of 2, 4 and 8
shards that all sit on device 0 (the kernels' work is the same in total; what differs is the
fan-out, the exchange rounds and the host merges).   ROWS=1000000 DIM=768 python tools/staged_shard_probe.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vettore_amd import nifs  # noqa: E402
from bench import build_shard, doc_ids  # noqa: E402


def main():
    n, d = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("DIM", 768))
    reps = int(os.environ.get("REPS", 300))
    dev = torch.device("cuda", 0)
    x = build_shard(torch, dev, n, d, 5)
    rng = np.random.default_rng(3)
    qs = rng.uniform(-1, 1, (16, d)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    base = None
    for shards in (0, 2, 4, 8):
        ref = nifs.flat_new_sharded(2, [0] * shards) if shards else nifs._flat_new(2)
        if shards:
            route = nifs.flat_route_ids(ref, doc_ids(0, n))
            for s in range(shards):
                idx = np.nonzero(route == s)[0]
                part = x[torch.from_numpy(idx).to(dev)].contiguous()
                assert nifs.flat_load_device_matrix(ref, doc_ids(0, 0, idx + 1), part.data_ptr(), len(idx), d) == ("ok", ())
                del part
        else:
            assert nifs.flat_load_device_matrix(ref, doc_ids(0, n), x.data_ptr(), n, d) == ("ok", ())
        calls = {
            "search": lambda q: nifs.flat_search(ref, q, 10),
            "quantized": lambda q: nifs.flat_quantized_search(ref, q, 100, 10),
            "funnel": lambda q: nifs.flat_funnel_search(ref, q, [128], 100, 10),
            "funnel2": lambda q: nifs.flat_funnel_search(ref, q, [64, 256], 100, 10),
            "hybrid": lambda q: nifs.flat_hybrid_search(ref, q, [(nifs.GEN_FUNNEL, 100, [128]), (nifs.GEN_QUANTIZED, 100, None), (nifs.GEN_SEARCH, 50, None)], 10),
        }
        row = {"shards": shards, "rows": n, "dim": d}
        answers = {}
        for name, fn in calls.items():
            for q in qs[:4]:
                fn(q)
            t0 = time.perf_counter()
            for i in range(reps):
                r = fn(qs[i % 16])
            row[name + "_us"] = round((time.perf_counter() - t0) / reps * 1e6, 1)
            assert r[0] == "ok", r
            answers[name] = [[(h[0], np.float32(h[1]).tobytes()) for h in fn(q)[1]] for q in qs[:4]]
        if base is None:
            base = answers
        row["equal_to_plain"] = answers == base
        print(json.dumps(row), flush=True)
        del ref


if __name__ == "__main__":
    main()
