#!/usr/bin/env python3
"""Single-row insert / upsert / delete rates through the C ABI (the reference's put/delete are
host-only and take microseconds).  Diagnostic only."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from vettore_amd import nifs, _lib  # noqa: E402

L = _lib.load()


def main():
    d, n = 768, 20000
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    ids = [b"doc-%08d" % i for i in range(n)]
    ref = nifs.flat_new_cosine()
    t0 = time.perf_counter()
    for i in range(n):
        assert L.vt_flat_insert(ref.handle, ids[i], len(ids[i]), x[i].ctypes.data_as(C.POINTER(C.c_float)), d) == 0
    t1 = time.perf_counter()
    print(json.dumps({"op": "insert (sorted ids)", "per_call_us": round((t1 - t0) / n * 1e6, 2)}), flush=True)
    t0 = time.perf_counter()
    for i in range(0, n, 4):
        assert L.vt_flat_insert(ref.handle, ids[i], len(ids[i]), x[(i + 1) % n].ctypes.data_as(C.POINTER(C.c_float)), d) == 0
    t1 = time.perf_counter()
    print(json.dumps({"op": "upsert", "per_call_us": round((t1 - t0) / (n // 4) * 1e6, 2)}), flush=True)
    h = C.c_void_p()
    t0 = time.perf_counter()
    for i in range(200):
        assert L.vt_flat_insert(ref.handle, b"zz-%d" % i, len(b"zz-%d" % i), x[i].ctypes.data_as(C.POINTER(C.c_float)), d) == 0
        assert L.vt_flat_search(ref.handle, x[i].ctypes.data_as(C.POINTER(C.c_float)), d, 3, C.byref(h)) == 0
        L.vt_hits_free(h)
    t1 = time.perf_counter()
    print(json.dumps({"op": "insert + search alternating", "per_pair_us": round((t1 - t0) / 200 * 1e6, 2)}), flush=True)
    t0 = time.perf_counter()
    for i in range(0, n, 5):
        assert L.vt_flat_delete(ref.handle, ids[i], len(ids[i])) == 0
    t1 = time.perf_counter()
    print(json.dumps({"op": "delete", "per_call_us": round((t1 - t0) / (n // 5) * 1e6, 2)}), flush=True)


if __name__ == "__main__":
    main()


def big():
    """unsorted inserts interleaved with searches on a 2M-row corpus (lazy rank path)"""
    d, n = 768, 2_000_000
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    ref = nifs.flat_new_cosine()
    assert nifs.flat_load_matrix(ref, ids, x) == ("ok", ())
    h = C.c_void_p()
    q = x[5].ctypes.data_as(C.POINTER(C.c_float))
    assert L.vt_flat_search(ref.handle, q, d, 10, C.byref(h)) == 0
    L.vt_hits_free(h)
    for label, env in (("lazy", None), ("eager", "1")):
        if env:
            nifs.debug_set("eager_ranks", int(env))
        t0 = time.perf_counter()
        for i in range(30):
            id_ = b"new-%s-%d" % (label.encode(), i)
            v = x[i + 100]
            assert L.vt_flat_insert(ref.handle, id_, len(id_), v.ctypes.data_as(C.POINTER(C.c_float)), d) == 0
            assert L.vt_flat_search(ref.handle, q, d, 10, C.byref(h)) == 0
            L.vt_hits_free(h)
        t1 = time.perf_counter()
        print(json.dumps({"op": "2M rows: unsorted insert + search, " + label, "per_pair_ms": round((t1 - t0) / 30 * 1e3, 3)}), flush=True)


if __name__ == "__main__" and os.environ.get("BIG"):
    big()
