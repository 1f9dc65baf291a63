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


def rows_f32(n, d, seed):
    """n x d uniform(-1, 1) floats, made in pieces (a 10 M x 768 sample is 30.7 GB: it must not exist in f64 as well)"""
    rng = np.random.default_rng(seed)
    x = np.empty((n, d), dtype=np.float32)
    for s0 in range(0, n, 1 << 17):
        e0 = min(n, s0 + (1 << 17))
        x[s0:e0] = rng.random((e0 - s0, d), dtype=np.float32) * 2.0 - 1.0
    return x


def main():
    n, d = int(os.environ.get("ROWS", "2000000")), int(os.environ.get("D", "768"))
    sorted_ids = os.environ.get("IDS", "sorted") == "sorted"
    x = rows_f32(n, d, 1)
    # (ids in bytewise order -- zero-padded, what a snapshot rebuild sorted by id hands over, collection.ex:427-433 -- or not)
    ids = [b"doc-%09d" % (i + 1) for i in range(n)] if sorted_ids else [b"doc-%d" % (i + 1) for i in range(n)]
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
    if os.environ.get("SLEEP_MS"):
        time.sleep(float(os.environ["SLEEP_MS"]) / 1e3)
        t1 = time.perf_counter()
    nifs.flat_set_profiling(ref, True)
    st, hits = nifs.flat_search(ref, x[5], 3)  # first search: id ranking happens here if it was deferred
    t2 = time.perf_counter()
    p0 = nifs.flat_get_profile(ref, reset=True)
    print(json.dumps({"first_search_kernel_ms": round(p0["scan_ms"], 3), "first_search_wall_ms": round((t2 - t1) * 1e3, 3)}), flush=True)
    nifs.flat_set_profiling(ref, False)
    assert hits[0][0] == ids[5], hits
    more = []
    for _ in range(3):
        ta = time.perf_counter()
        nifs.flat_search(ref, x[6], 3)
        more.append(round(time.perf_counter() - ta, 4))
    print(json.dumps({"following_searches_s": more}), flush=True)
    print(json.dumps({"op": "load_matrix", "ids": "sorted" if sorted_ids else "unsorted", "serial": os.environ.get("VT_INGEST_SERIAL") is not None,
                      "rows": n, "d": d, "seconds": round(t1 - t0, 3), "rows_per_s": round(n / (t1 - t0)),
                      "GBps": round(n * d * 4 / (t1 - t0) / 1e9, 2), "first_search_s": round(t2 - t1, 3)}), flush=True)
    if os.environ.get("SHIM"):
        # the same rows through the erl_nif shim's flat_load_binary (fake term runtime, tests/nif_runtime.py): what the
        # Elixir side's one-binary put_many costs on top (id list decoding, the binary is used in place)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import nif_runtime
        rt = nif_runtime.Runtime()
        ref2 = rt.call("flat_new", 2, [0])
        blob = x.tobytes()
        res, secs = rt.call_timed("flat_load_binary", ref2, ids, blob, d)
        print(json.dumps({"op": "shim flat_load_binary (fake erl_nif runtime; the NIF call alone: id list decoding + vt_flat_load_matrix on the binary in place)",
                          "result": repr(res), "seconds": round(secs, 3), "GBps": round(n * d * 4 / secs / 1e9, 2)}), flush=True)
        del ref2
    if os.environ.get("LOAD_ONLY"):
        return
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
