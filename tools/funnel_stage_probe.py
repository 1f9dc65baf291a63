#!/usr/bin/env python3
"""funnel_search latency per prefix length on a plain handle (ROWS, DIM, METRIC in the environment)."""
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
    n, d = int(os.environ.get("ROWS", 100_000)), int(os.environ.get("DIM", 768))
    metric = int(os.environ.get("METRIC", 2))
    reps = int(os.environ.get("REPS", 300))
    dev = torch.device("cuda", 0)
    x = build_shard(torch, dev, n, d, 5)
    ref = nifs._flat_new(metric)
    assert nifs.flat_load_device_matrix(ref, doc_ids(0, n), x.data_ptr(), n, d) == ("ok", ())
    q = np.random.default_rng(3).uniform(-1, 1, d).astype(np.float32)
    q /= np.linalg.norm(q)
    for stages in ([32], [64], [96], [128], [192], [256], [384], [768], [64, 256], [128, 256]):
        for cand in (100,):
            for _ in range(10):
                nifs.flat_funnel_search(ref, q, stages, cand, 10)
            nifs.flat_set_profiling(ref, True)
            nifs.flat_get_profile(ref, reset=True)
            t0 = time.perf_counter()
            for _ in range(reps):
                nifs.flat_funnel_search(ref, q, stages, cand, 10)
            us = (time.perf_counter() - t0) / reps * 1e6
            prof = nifs.flat_get_profile(ref, reset=True)
            nifs.flat_set_profiling(ref, False)
            print(json.dumps({"rows": n, "stages": stages, "candidates": cand, "us": round(us, 1),
                              "prefix_kernel_us": round(prof["prefix_ms"] / max(1, prof["prefix_launches"]) * 1e3, 1),
                              "prefix_launches": prof["prefix_launches"] / reps}), flush=True)


if __name__ == "__main__":
    main()
