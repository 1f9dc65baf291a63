#!/usr/bin/env python3
"""K1p (and, under cosine, K6bm: "cs_panel") on 64-float against 32-float panels (vt_debug_set "pm_panel"), ALTERNATING in one process on one corpus: ms per
sweep of eight queries, per metric and prefix length (boxes of the pool differ by more than the two builds do: only an
alternating run says which is ahead).  ROWS / DIM / METRICS / PREFIXES / ROUNDS env.  Diagnostic only."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vettore_amd import nifs, _lib  # noqa: E402
from bench import build_shard, doc_ids  # noqa: E402

L = _lib.load()


def main():
    rows = int(os.environ.get("ROWS", 10_000_000))
    dim = int(os.environ.get("DIM", 768))
    prefixes = [int(v) for v in os.environ.get("PREFIXES", "64,128,256,%d" % dim).split(",")]
    rounds = int(os.environ.get("ROUNDS", 5))
    nq = int(os.environ.get("NQ", 8))   # (NQ=1: the single-query kernels -- K6b under cosine, K1 on the prefix otherwise)
    nifs.debug_set("batch_no_mfma", 1)
    rng = np.random.default_rng(0)
    x = build_shard(torch, torch.device("cuda", 0), rows, dim, 99)
    for metric in (int(m) for m in os.environ.get("METRICS", "3,0,5").split(",")):
        ref = nifs._flat_new(metric)
        assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
        nifs.flat_set_profiling(ref, True)
        qs = rng.uniform(-1, 1, (8, dim)).astype(np.float32)
        qsp = qs.ctypes.data_as(C.POINTER(C.c_float))
        outs = (C.c_void_p * 8)()
        out = {"metric": nifs.METRICS[metric], "rows": rows, "dim": dim}
        for d1 in prefixes:
            st = (C.c_size_t * 1)(d1)

            def sweep_ms(panel, blocks):
                nifs.debug_set("cs_panel" if metric == 2 else "pm_panel", panel)
                nifs.debug_set("pm_blocks", blocks)
                def call():
                    if nq == 1:
                        assert L.vt_flat_funnel_search(ref.handle, qsp, dim, st, 1, 100, 10, outs) == 0
                    else:
                        assert L.vt_flat_funnel_search_batch(ref.handle, qsp, nq, dim, st, 1, 100, 10, outs) == 0
                    L.vt_hits_free_many(outs, nq)
                call()
                nifs.flat_get_profile(ref, reset=True)
                for _ in range(6):
                    call()
                p = nifs.flat_get_profile(ref, reset=True)
                # (the sample pass of every call is a launch of its own: half the launches, a few tens of us each)
                return p["prefix_ms"] / max(1, p["prefix_launches"])
            res = {"64x2": [], "32x2": []}
            for _ in range(rounds):
                res["64x2"].append(sweep_ms(64, 2))
                res["32x2"].append(sweep_ms(32, 2))
            out["prefix%d" % d1] = {k: {"min": round(min(v), 4), "median": round(sorted(v)[len(v) // 2], 4)} for k, v in res.items()}
        nifs.debug_set("pm_panel", 0)
        nifs.debug_set("cs_panel", 0)
        nifs.debug_set("pm_blocks", 0)
        print(json.dumps(out), flush=True)
        del ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
