#!/usr/bin/env python3
"""Per-call time of the main entry points across corpus sizes (looking for cliffs at the
switch points between code paths: tile heights, K4/K4h at 16 384 rows, threshold select at
65 536 rows, MFMA batches at 4 096 rows).  Diagnostic only."""
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


def timeit(fn, reps=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    dim = 768
    rng = np.random.default_rng(0)
    for rows in (1000, 4000, 4200, 16000, 17000, 60000, 70000, 130000, 140000, 260000, 270000, 1000000):
        x = build_shard(torch, torch.device("cuda", 0), rows, dim, 99)
        ref = nifs._flat_new(2)
        assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
        del x
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        q /= np.linalg.norm(q)
        qp = q.ctypes.data_as(C.POINTER(C.c_float))
        h = C.c_void_p()
        qs = rng.uniform(-1, 1, (16, dim)).astype(np.float32)
        qsp = qs.ctypes.data_as(C.POINTER(C.c_float))
        outs = (C.c_void_p * 16)()
        st = (C.c_size_t * 1)(128)

        def search(k):
            assert L.vt_flat_search(ref.handle, qp, dim, k, C.byref(h)) == 0
            L.vt_hits_free(h)

        def quant(c):
            assert L.vt_flat_quantized_search(ref.handle, qp, dim, c, 10, C.byref(h)) == 0
            L.vt_hits_free(h)

        def funnel(c):
            assert L.vt_flat_funnel_search(ref.handle, qp, dim, st, 1, c, 10, C.byref(h)) == 0
            L.vt_hits_free(h)

        def batch():
            assert L.vt_flat_search_batch(ref.handle, qsp, 16, dim, 10, outs) == 0
            for i in range(16):
                L.vt_hits_free(C.c_void_p(outs[i]))

        out = {"rows": rows,
               "search10_us": round(timeit(lambda: search(10)), 1),
               "search300_us": round(timeit(lambda: search(300)), 1),
               "quant100_us": round(timeit(lambda: quant(100)), 1),
               "quant1000_us": round(timeit(lambda: quant(1000)), 1),
               "funnel100_us": round(timeit(lambda: funnel(100)), 1),
               "funnel1000_us": round(timeit(lambda: funnel(1000)), 1),
               "batch16_us": round(timeit(batch, 5), 1)}
        print(json.dumps(out), flush=True)
        del ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
