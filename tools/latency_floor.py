#!/usr/bin/env python3
"""Fixed per-call cost of the C ABI entry points on one MI355X: tiny corpora, so the
kernels are near-empty and what is left is upload + launches + wait.  Diagnostic only."""
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


def main():
    L = _lib.load()
    dev = torch.device("cuda", 0)
    sizes = [int(v) for v in sys.argv[1:]] or [4096, 1_000_000]
    for rows in sizes:
        dim = 768
        x = build_shard(torch, dev, rows, dim, 1234)
        ref = nifs._flat_new(2)
        assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
        rng = np.random.default_rng(1)
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        q /= np.linalg.norm(q)
        qp = q.ctypes.data_as(C.POINTER(C.c_float))
        h = C.c_void_p()
        st = (C.c_size_t * 2)(128, 256)

        def search():
            assert L.vt_flat_search(ref.handle, qp, dim, 10, C.byref(h)) == 0
            L.vt_hits_free(h)

        def quant():
            assert L.vt_flat_quantized_search(ref.handle, qp, dim, 100, 10, C.byref(h)) == 0
            L.vt_hits_free(h)

        def funnel():
            assert L.vt_flat_funnel_search(ref.handle, qp, dim, st, 2, 100, 10, C.byref(h)) == 0
            L.vt_hits_free(h)

        for name, fn in (("search", search), ("quantized", quant), ("funnel", funnel)):
            for _ in range(20):
                fn()
            n = 300
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            dt = (time.perf_counter() - t0) / n
            print(json.dumps({"rows": rows, "call": name, "us_per_call": round(dt * 1e6, 1)}), flush=True)
        del ref, x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
