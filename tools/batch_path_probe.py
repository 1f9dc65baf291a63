#!/usr/bin/env python3
"""Which way a batch should go: for a few row widths and batch sizes, the time of flat_search_batch
as the library chooses (default) against each way forced -- the shared matrix-core pass (K2), the
multi-query sweep (K1m), one scan per query.  `default` should sit on the minimum of the three.
    BYTES=1500000000 DIMS=128,256,384,768 python tools/batch_path_probe.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vettore_amd import nifs  # noqa: E402
from bench import build_shard, doc_ids  # noqa: E402

WAYS = {
    "default": {},
    "k2": {"no_multi_scan": 1},
    "k1m": {"batch_no_mfma": 1},
    "singles": {"no_multi_scan": 1, "batch_no_mfma": 1},
}


def main():
    total = float(os.environ.get("BYTES", 1.5e9))
    dims = [int(v) for v in os.environ.get("DIMS", "128,256,384,768").split(",")]
    metrics = [int(v) for v in os.environ.get("METRICS", "2").split(",")]
    sizes = [int(v) for v in os.environ.get("NQS", "2,4,8,16,32,64").split(",")]
    dev = torch.device("cuda", 0)
    for d in dims:
        n = int(total / (d * 4))
        x = build_shard(torch, dev, n, d, 5)
        for metric in metrics:
            ref = nifs._flat_new(metric)
            assert nifs.flat_load_device_matrix(ref, doc_ids(0, n), x.data_ptr(), n, d) == ("ok", ())
            rng = np.random.default_rng(3)
            for nq in sizes:
                qs = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
                qs /= np.linalg.norm(qs, axis=1, keepdims=True)
                row = {"d": d, "rows": n, "metric": metric, "nq": nq}
                for way, env in WAYS.items():
                    for k in ("no_multi_scan", "batch_no_mfma"):   # (vt_debug_set: the environment is read once, at load)
                        nifs.debug_set(k, env.get(k, 0))
                    for _ in range(3):
                        r = nifs.flat_search_batch(ref, qs, 10)
                    assert r[0] == "ok"
                    reps = 10
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        nifs.flat_search_batch(ref, qs, 10)
                    row[way + "_us"] = round((time.perf_counter() - t0) / reps * 1e6)
                for k in ("no_multi_scan", "batch_no_mfma"):
                    nifs.debug_set(k, 0)
                best = min(row[w + "_us"] for w in ("k2", "k1m", "singles"))
                row["default_over_best"] = round(row["default_us"] / best, 2)
                print(json.dumps(row), flush=True)
            del ref
        del x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
