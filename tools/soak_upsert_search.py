#!/usr/bin/env python3
"""Soak test: rewrite a row (MODE=upsert: in place; MODE=move: delete + insert, so that it moves to
the end of the slab), then search for exactly that vector (L2, limit 1): the answer must be that id
at distance 0.  Deterministic per seed; ~14 000 iterations per second.
    SECONDS=150 MODE=move python tools/soak_upsert_search.py
"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from vettore_amd import nifs, _lib
L = _lib.load()
n, d = int(os.environ.get("N", 600)), int(os.environ.get("D", 16))
rng = np.random.default_rng(1)
x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
ids = [b"id-%d" % i for i in range(n)]
ref = nifs._flat_new(0)
assert nifs.flat_load_matrix(ref, ids, x) == ("ok", ())
h = C.c_void_p()
fp = C.POINTER(C.c_float)
bad = it = 0
t0 = time.time()
budget = float(os.environ.get("SECONDS", 100))
stop_after = int(os.environ.get("STOP_AFTER", 10**12))
mode = os.environ.get("MODE", "upsert")
while time.time() - t0 < budget:
    if it >= stop_after:
        break
    for _ in range(2000):
        it += 1
        r = int(rng.integers(0, n))
        v = rng.uniform(-1, 1, d).astype(np.float32)
        if mode == "upsert":
            assert L.vt_flat_insert(ref.handle, ids[r], len(ids[r]), v.ctypes.data_as(fp), d) == 0
        else:  # delete + insert: the row moves
            assert L.vt_flat_delete(ref.handle, ids[r], len(ids[r])) == 0
            assert L.vt_flat_insert(ref.handle, ids[r], len(ids[r]), v.ctypes.data_as(fp), d) == 0
        assert L.vt_flat_search(ref.handle, v.ctypes.data_as(fp), d, 1, C.byref(h)) == 0
        ln = C.c_size_t()
        got = C.string_at(L.vt_hits_id(h, 0, C.byref(ln)), ln.value)
        raw = L.vt_hits_raw(h, 0)
        L.vt_hits_free(h)
        if got != ids[r] or raw != 0.0:
            bad += 1
            if bad <= 5:
                assert L.vt_flat_search(ref.handle, v.ctypes.data_as(fp), d, 1, C.byref(h)) == 0
                got2 = C.string_at(L.vt_hits_id(h, 0, C.byref(ln)), ln.value); raw2 = L.vt_hits_raw(h, 0)
                L.vt_hits_free(h)
                print("STALE it", it, "row", r, "want", ids[r], "got", got, raw, "again", got2, raw2, "n", len(ref), flush=True)
                for kk in (2, 3, 10, 100, 300, 600):
                    res = nifs.flat_search(ref, v, kk)[1]
                    pos = [i for i, hh in enumerate(res) if hh[0] == ids[r]]
                    print("   k", kk, "first", res[0], "pos of wanted", pos, flush=True)
                res = nifs.flat_search(ref, v, 1)[1]
                print("   k 1 after", res, flush=True)
print("mode", mode, "iterations", it, "bad", bad, "per s", round(it / (time.time() - t0)))
