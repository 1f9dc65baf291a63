#!/usr/bin/env python3
"""Looks for performance cliffs off the headline configurations: odd batch sizes, limits
around the buffer-size switches, awkward dimensions, candidate counts on both sides of the
device-chained paths.  Prints one line per case: e2e ms and the ratio to the plain scan of
the same corpus.  Diagnostic only."""
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


def make(metric, rows, dim):
    x = build_shard(torch, torch.device("cuda", 0), rows, dim, 99)
    ref = nifs._flat_new(metric)
    assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
    del x
    torch.cuda.empty_cache()
    return ref


def timeit(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    rng = np.random.default_rng(0)
    rows, dim = 2_000_000, 768
    for metric in (2, 0):
        ref = make(metric, rows, dim)
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        q /= np.linalg.norm(q)
        qp = q.ctypes.data_as(C.POINTER(C.c_float))
        h = C.c_void_p()

        def search(k):
            assert L.vt_flat_search(ref.handle, qp, dim, k, C.byref(h)) == 0
            L.vt_hits_free(h)

        base = timeit(lambda: search(10))
        print(json.dumps({"metric": metric, "case": "search k=10", "ms": round(base, 3)}), flush=True)
        for k in (1, 64, 65, 256, 257, 512):
            ms = timeit(lambda: search(k))
            print(json.dumps({"metric": metric, "case": "search k=%d" % k, "ms": round(ms, 3), "x_base": round(ms / base, 2)}), flush=True)
        for nq in (2, 3, 33, 100, 257, 300):
            qs = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
            outs = (C.c_void_p * nq)()
            qsp = qs.ctypes.data_as(C.POINTER(C.c_float))

            def batch():
                assert L.vt_flat_search_batch(ref.handle, qsp, nq, dim, 10, outs) == 0
                for i in range(nq):
                    L.vt_hits_free(C.c_void_p(outs[i]))

            ms = timeit(batch, 3)
            print(json.dumps({"metric": metric, "case": "batch nq=%d" % nq, "ms": round(ms, 3), "x_base": round(ms / base, 2),
                              "ms_per_query": round(ms / nq, 3)}), flush=True)
        for cand in (1, 10, 256, 257, 1000):
            def quant():
                assert L.vt_flat_quantized_search(ref.handle, qp, dim, cand, min(cand, 10), C.byref(h)) == 0
                L.vt_hits_free(h)

            ms = timeit(quant)
            print(json.dumps({"metric": metric, "case": "quantized candidates=%d" % cand, "ms": round(ms, 3), "x_base": round(ms / base, 2)}), flush=True)
        st = (C.c_size_t * 2)(64, 256)
        for cand in (100, 256, 300):
            def funnel():
                assert L.vt_flat_funnel_search(ref.handle, qp, dim, st, 2, cand, 10, C.byref(h)) == 0
                L.vt_hits_free(h)

            ms = timeit(funnel)
            print(json.dumps({"metric": metric, "case": "funnel [64,256] candidates=%d" % cand, "ms": round(ms, 3), "x_base": round(ms / base, 2)}), flush=True)
        del ref
    for dim in (64, 65, 767, 769, 3072, 4096, 8192):
        rows = max(50_000, 1_500_000_000 // (((dim + 63) // 64) * 64 * 4))
        ref = make(2, rows, dim)
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        qp = q.ctypes.data_as(C.POINTER(C.c_float))
        h = C.c_void_p()

        def search():
            assert L.vt_flat_search(ref.handle, qp, dim, 10, C.byref(h)) == 0
            L.vt_hits_free(h)

        ms = timeit(search)
        padded = ((dim + 63) // 64) * 64
        print(json.dumps({"case": "dim=%d rows=%d" % (dim, rows), "ms": round(ms, 3),
                          "GBps_useful": round(rows * dim * 4 / ms / 1e6, 1), "GBps_padded": round(rows * padded * 4 / ms / 1e6, 1)}), flush=True)
        del ref


if __name__ == "__main__":
    main()
