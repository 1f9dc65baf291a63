#!/usr/bin/env python3
"""Per-call time of every entry point under each of the nine metrics (one mid-size corpus):
looks for a metric that falls off a fast path.  Diagnostic only."""
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


def timeit(fn, reps=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return round((time.perf_counter() - t0) / reps * 1e6, 1)


def main():
    rows, dim = 500_000, 256
    rng = np.random.default_rng(0)
    for metric in range(9):
        x = build_shard(torch, torch.device("cuda", 0), rows, dim, 99)
        ref = nifs._flat_new(metric)
        assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
        del x
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        qp = q.ctypes.data_as(C.POINTER(C.c_float))
        h = C.c_void_p()
        qs = rng.uniform(-1, 1, (16, dim)).astype(np.float32)
        qsp = qs.ctypes.data_as(C.POINTER(C.c_float))
        outs = (C.c_void_p * 16)()
        st = (C.c_size_t * 1)(64)

        def search(k):
            assert L.vt_flat_search(ref.handle, qp, dim, k, C.byref(h)) == 0
            L.vt_hits_free(h)

        def quant():
            assert L.vt_flat_quantized_search(ref.handle, qp, dim, 100, 10, C.byref(h)) == 0
            L.vt_hits_free(h)

        def funnel():
            assert L.vt_flat_funnel_search(ref.handle, qp, dim, st, 1, 100, 10, C.byref(h)) == 0
            L.vt_hits_free(h)

        def batch():
            assert L.vt_flat_search_batch(ref.handle, qsp, 16, dim, 10, outs) == 0
            for i in range(16):
                L.vt_hits_free(C.c_void_p(outs[i]))

        print(json.dumps({"metric": nifs.METRICS[metric], "search10_us": timeit(lambda: search(10)),
                          "search1000_us": timeit(lambda: search(1000), 3), "quantized_us": timeit(quant),
                          "funnel_us": timeit(funnel), "batch16_us": timeit(batch, 3)}), flush=True)
        del ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
