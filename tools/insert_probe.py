#!/usr/bin/env python3
"""What one mutation costs through the C ABI: vt_flat_insert of new ids one by one (flat.rs:60-74; Vettore.put/2 ->
Index.Flat.put/2 -> one NIF call per record), upserts of existing ids, deletes, and vt_flat_insert_many in blocks of
100 / 10 000 -- on an index that already holds ROWS rows (default 1 M x 768).  Diagnostic only."""
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
    rows = int(os.environ.get("ROWS", 1_000_000))
    dim = int(os.environ.get("DIM", 768))
    x = build_shard(torch, torch.device("cuda", 0), rows, dim, 7)
    ref = nifs._flat_new(2)
    assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
    del x
    rng = np.random.default_rng(3)
    n = 5000
    vecs = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    out = {"rows": rows, "dim": dim}
    ids = [b"new-%07d" % i for i in range(n)]
    # (the first append behind a bulk load doubles the id table, the id vector and the host rank column -- 80 ms for a
    # million ids, once: not what a steady trickle pays)
    warm = rng.uniform(-1, 1, dim).astype(np.float32)
    assert L.vt_flat_insert(ref.handle, b"warm", 4, warm.ctypes.data_as(C.POINTER(C.c_float)), dim) == 0
    t0 = time.perf_counter()
    for i in range(n):
        assert L.vt_flat_insert(ref.handle, ids[i], len(ids[i]), vecs[i].ctypes.data_as(C.POINTER(C.c_float)), dim) == 0
    out["insert_new_us"] = round((time.perf_counter() - t0) / n * 1e6, 2)
    t0 = time.perf_counter()
    for i in range(n):
        assert L.vt_flat_insert(ref.handle, ids[i], len(ids[i]), vecs[n - 1 - i].ctypes.data_as(C.POINTER(C.c_float)), dim) == 0
    out["upsert_us"] = round((time.perf_counter() - t0) / n * 1e6, 2)
    q = vecs[0]
    t0 = time.perf_counter()
    for i in range(200):   # a search between inserts: the id ranks are brought up to date lazily
        assert L.vt_flat_insert(ref.handle, ids[i], len(ids[i]), vecs[i].ctypes.data_as(C.POINTER(C.c_float)), dim) == 0
        h = C.c_void_p()
        assert L.vt_flat_search(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), dim, 10, C.byref(h)) == 0
        L.vt_hits_free(h)
    out["insert_then_search_us"] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
    t0 = time.perf_counter()
    for i in range(n):
        assert L.vt_flat_delete(ref.handle, ids[i], len(ids[i])) == 0
    out["delete_us"] = round((time.perf_counter() - t0) / n * 1e6, 2)
    for block in (100, 10_000):
        m = 20_000
        v = rng.uniform(-1, 1, (m, dim)).astype(np.float32)
        bid = [b"blk%d-%07d" % (block, i) for i in range(m)]
        t0 = time.perf_counter()
        for lo in range(0, m, block):
            res = nifs.flat_insert_many(ref, list(zip(bid[lo:lo + block], v[lo:lo + block])))
            assert res[0] == "ok", res
        out["insert_many_%d_us_per_row" % block] = round((time.perf_counter() - t0) / m * 1e6, 2)
    # The figures above time the Python loop as r04 / r05 did (a ctypes pointer conversion per call rides along).  The C ABI
    # call alone, pointers prepared beforehand, per call: a tight loop runs at the speed the device drains the landing ring
    # (one small kernel and an event per mutation), a lone call returns at about the median.
    ptrs = [vecs[i].ctypes.data_as(C.POINTER(C.c_float)) for i in range(n)]
    ids2 = [b"again-%07d" % i for i in range(n)]
    for name, keys in (("abi_insert_new", ids2), ("abi_upsert", ids2), ("abi_delete", ids2)):
        t = np.empty(n)
        t_all = time.perf_counter()
        for i in range(n):
            t0 = time.perf_counter()
            if name == "abi_delete":
                st = L.vt_flat_delete(ref.handle, keys[i], len(keys[i]))
            else:
                st = L.vt_flat_insert(ref.handle, keys[i], len(keys[i]), ptrs[i], dim)
            t[i] = time.perf_counter() - t0
            assert st == 0
        out[name] = {"mean_us": round((time.perf_counter() - t_all) / n * 1e6, 2), "p10_us": round(float(np.percentile(t, 10)) * 1e6, 2),
                     "p50_us": round(float(np.percentile(t, 50)) * 1e6, 2), "p90_us": round(float(np.percentile(t, 90)) * 1e6, 2),
                     "p99_us": round(float(np.percentile(t, 99)) * 1e6, 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
