#!/usr/bin/env python3
"""Ingest throughput of the C ABI: vt_flat_load_matrix from host memory (ids + rows),
vt_flat_insert_many in batches, deletes.  Diagnostic only."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from vettore_amd import nifs  # noqa: E402


def main():
    n, d = int(os.environ.get("ROWS", "2000000")), 768
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    ref = nifs.flat_new_cosine()
    tp = time.perf_counter()
    idb, ioff = nifs.pack_ids(ids)
    t0 = time.perf_counter()
    from vettore_amd import _lib
    L = _lib.load()
    rc = L.vt_flat_load_matrix(ref.handle, n, d, idb, ioff.ctypes.data_as(C.POINTER(C.c_size_t)),
                               x.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0
    t1 = time.perf_counter()
    print(json.dumps({"python_pack_ids_s": round(t0 - tp, 3)}), flush=True)
    st, hits = nifs.flat_search(ref, x[5], 3)  # first search: id ranking happens here if it was deferred
    t2 = time.perf_counter()
    print(json.dumps({"op": "load_matrix", "rows": n, "seconds": round(t1 - t0, 3), "rows_per_s": round(n / (t1 - t0)),
                      "GBps": round(n * d * 4 / (t1 - t0) / 1e9, 2), "first_search_s": round(t2 - t1, 3)}), flush=True)
    # incremental: 64 batches of 1000 rows with fresh ids, then a search (re-rank of the newcomers)
    extra = rng.uniform(-1, 1, size=(64000, d)).astype(np.float32)
    t0 = time.perf_counter()
    for b in range(64):
        items = [(b"new-%d" % (b * 1000 + i), extra[b * 1000 + i]) for i in range(1000)]
        assert nifs.flat_insert_many(ref, items) == ("ok", ())
    t1 = time.perf_counter()
    nifs.flat_search(ref, x[5], 3)
    t2 = time.perf_counter()
    print(json.dumps({"op": "insert_many 64 x 1000", "seconds": round(t1 - t0, 3), "rows_per_s": round(64000 / (t1 - t0)),
                      "search_after_s": round(t2 - t1, 4)}), flush=True)
    t0 = time.perf_counter()
    for i in range(2000):
        nifs.flat_delete(ref, b"doc-%d" % (i * 7 + 1))
    t1 = time.perf_counter()
    nifs.flat_search(ref, x[5], 3)
    t2 = time.perf_counter()
    print(json.dumps({"op": "delete x 2000", "seconds": round(t1 - t0, 3), "per_delete_us": round((t1 - t0) / 2000 * 1e6, 1),
                      "search_after_s": round(t2 - t1, 4)}), flush=True)


if __name__ == "__main__":
    main()
