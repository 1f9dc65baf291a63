#!/usr/bin/env python3
"""K1p (prefix_multi_kernel: K1's arithmetic for up to eight queries per sweep of the rows' prefixes) beside the
single-query prefix scan and -- at full width -- beside K1m: time per sweep, bytes of prefix per second.
ROWS / DIM / METRICS (codes) / PREFIXES env.  Diagnostic only."""
import ctypes as C
import json
import os
import sys
import time

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
            rec = {}
            for nq in (1, 8):
                def call():
                    if nq == 1:
                        assert L.vt_flat_funnel_search(ref.handle, qsp, dim, st, 1, 100, 10, outs) == 0
                    else:
                        assert L.vt_flat_funnel_search_batch(ref.handle, qsp, nq, dim, st, 1, 100, 10, outs) == 0
                    for i in range(nq):
                        L.vt_hits_free(C.c_void_p(outs[i]))
                call()
                nifs.flat_get_profile(ref, reset=True)
                reps = 10
                t0 = time.perf_counter()
                for _ in range(reps):
                    call()
                dt = (time.perf_counter() - t0) / reps
                p = nifs.flat_get_profile(ref, reset=True)
                # (a stage over the whole row is priced as a scan by the single path)
                ms = (p["prefix_ms"] + (p["scan_ms"] if d1 == dim and nq == 1 else 0.0)) / max(1, p["prefix_launches"] + (p["scan_launches"] if d1 == dim and nq == 1 else 0))
                rec["nq%d" % nq] = {"call_ms": round(dt * 1e3, 3), "sweep_ms": round(ms, 4), "GBps": round(rows * d1 * 4 / ms / 1e6, 0) if ms > 0 else None,
                                    "grouped": p["prefix_queries"] // reps}
            out["prefix%d" % d1] = rec
        if os.environ.get("K1M", "1") == "1":
            def call():
                assert L.vt_flat_search_batch(ref.handle, qsp, 8, dim, 10, outs) == 0
                for i in range(8):
                    L.vt_hits_free(C.c_void_p(outs[i]))
            call()
            nifs.flat_get_profile(ref, reset=True)
            for _ in range(10):
                call()
            p = nifs.flat_get_profile(ref, reset=True)
            ms = p["scan_ms"] / max(1, p["scan_launches"])
            out["k1m_nq8"] = {"sweep_ms": round(ms, 4), "GBps": round(rows * dim * 4 / ms / 1e6, 0)}
        print(json.dumps(out), flush=True)
        del ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
