#!/usr/bin/env python3
"""The groups of one call over two contexts (r05, funnel_groups): funnel_search batches (stage-1 sweeps of eight: K6bm
under cosine, K1p under L2), plain search batches that run as K1p sweeps (manhattan) and quantized_search batches (K4hm), with the pipeline on and off
alternating in ONE process on one index (`no_group_pipeline` by name).  Every batched list is compared with the same
query's single call.  ROWS / DIM / NQS / REPS / LEGS (funnel,search,quantized) env.  Diagnostic only."""
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
from bench import build_shard, doc_ids, hits_of  # noqa: E402

L = _lib.load()


def main():
    rows = int(os.environ.get("ROWS", 10_000_000))
    dim = int(os.environ.get("DIM", 768))
    reps = int(os.environ.get("REPS", 6))
    nqs = [int(v) for v in os.environ.get("NQS", "16,64,256").split(",")]
    rng = np.random.default_rng(5)
    st_arr = (C.c_size_t * 1)(min(dim, 128))
    legs = (("funnel cosine", 2, "funnel"), ("funnel l2", 0, "funnel"), ("search manhattan", 5, "search"),
            ("quantized cosine", 2, "quantized"))
    only = os.environ.get("LEGS")
    for name, metric, kind in legs:
        if only and name.split()[0] not in only.split(","):
            continue
        funnel = kind == "funnel"
        x = build_shard(torch, torch.device("cuda", 0), rows, dim, 99, normalize=(metric == 2))
        ref = nifs._flat_new(metric)
        assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
        del x
        torch.cuda.empty_cache()
        out = {"leg": name, "rows": rows, "dim": dim}
        for nq in nqs:
            if kind == "search" and nq > 64:
                continue
            qs = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
            if metric == 2:
                qs /= np.linalg.norm(qs.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
            qsp = qs.ctypes.data_as(C.POINTER(C.c_float))
            outs = (C.c_void_p * nq)()

            def call(keep=False):
                if funnel:
                    st = L.vt_flat_funnel_search_batch(ref.handle, qsp, nq, dim, st_arr, 1, 100, 10, outs)
                elif kind == "quantized":
                    st = L.vt_flat_quantized_search_batch(ref.handle, qsp, nq, dim, 100, 10, outs)
                else:
                    st = L.vt_flat_search_batch(ref.handle, qsp, nq, dim, 10, outs)
                assert st == 0, (L.vt_last_error() or b"").decode()
                if keep:
                    return [hits_of(L, C.c_void_p(outs[i])) for i in range(nq)]
                L.vt_hits_free_many(outs, nq)

            def single(i):
                h = C.c_void_p()
                qp = qs[i].ctypes.data_as(C.POINTER(C.c_float))
                if funnel:
                    assert L.vt_flat_funnel_search(ref.handle, qp, dim, st_arr, 1, 100, 10, C.byref(h)) == 0
                elif kind == "quantized":
                    assert L.vt_flat_quantized_search(ref.handle, qp, dim, 100, 10, C.byref(h)) == 0
                else:
                    assert L.vt_flat_search(ref.handle, qp, dim, 10, C.byref(h)) == 0
                return hits_of(L, h)

            res = {}
            for off in (1, 0, 1, 0):
                nifs.debug_set("no_group_pipeline", off)
                call()
                t0 = time.perf_counter()
                for _ in range(reps):
                    call()
                dt = (time.perf_counter() - t0) / reps
                key = "series" if off else "pipelined"
                res[key] = min(res.get(key, 1e9), dt * 1e3)
            nifs.debug_set("no_group_pipeline", 0)
            got = call(keep=True)
            for i in (0, 7, 8, nq // 2, nq - 1):
                assert got[i] == single(i), (name, nq, i)
            out["nq%d" % nq] = {"series_ms": round(res["series"], 3), "pipelined_ms": round(res["pipelined"], 3),
                                "queries_per_s": round(nq / res["pipelined"] * 1e3), "was": round(nq / res["series"] * 1e3)}
            if "dense_sample_ms" in res:
                out["nq%d" % nq]["dense_sample_ms"] = round(res["dense_sample_ms"], 3)
            if "one_stream" in res:
                out["nq%d" % nq]["one_stream_ms"] = round(res["one_stream"], 3)
        print(json.dumps(out), flush=True)
        del ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
