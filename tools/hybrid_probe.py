#!/usr/bin/env python3
"""tools/hybrid_probe.py -- hybrid_search (collection.ex:325-345) per call: the one-wait device chain
(VT_HYBRID_CHAIN=1) against the default host-composed path, same index, same queries.

  ROWS=1000000 DIM=768 python tools/hybrid_probe.py        # one JSON line per mode

Generators: funnel (stages d/6, d/3; 100 candidates), quantized (100), the index's own search (100);
limit 10; cosine.  Diagnostic only."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vettore_amd import nifs  # noqa: E402
from bench import build_shard, doc_ids, normalized_queries  # noqa: E402


def main():
    rows = int(os.environ.get("ROWS", 1_000_000))
    dim = int(os.environ.get("DIM", 768))
    calls = int(os.environ.get("CALLS", 300))
    metric = int(os.environ.get("METRIC", 2))
    x = build_shard(torch, torch.device("cuda", 0), rows, dim, 99)
    ref = nifs._flat_new(metric)
    assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
    del x
    qs = normalized_queries(calls + 20, dim, 5)
    gens = [(nifs.GEN_FUNNEL, 100, [dim // 6, dim // 3]), (nifs.GEN_QUANTIZED, 100, []), (nifs.GEN_SEARCH, 100, [])]
    results, total = {}, {"chain": 0.0, "host": 0.0}
    rounds = 4  # the two modes alternate, so that clock drift and warm-up fall on both
    per = calls // rounds
    for r in range(rounds + 1):
        for mode in ("chain", "host"):
            nifs.debug_set("hybrid_chain", 1 if mode == "chain" else 0)
            t0 = time.perf_counter()
            for q in qs[20 + (r % rounds) * per: 20 + (r % rounds) * per + per]:
                results[mode] = nifs.flat_hybrid_search(ref, q, gens, 10)
            if r:  # (round 0 warms up)
                total[mode] += time.perf_counter() - t0
    for mode in ("chain", "host"):
        print(json.dumps({"mode": mode, "metric": nifs.METRICS[metric], "rows": rows, "dim": dim, "calls": per * rounds,
                          "ms_per_call": round(total[mode] / (per * rounds) * 1e3, 4)}), flush=True)
    assert results["chain"] == results["host"]


if __name__ == "__main__":
    main()
