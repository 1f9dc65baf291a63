import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vettore_amd import nifs, _lib
from bench import build_shard, doc_ids, normalized_queries
L = _lib.load()
rows, dim = int(os.environ.get("ROWS", 1_000_000)), 768
x = build_shard(torch, torch.device("cuda", 0), rows, dim, 5)
ref = nifs.flat_new_cosine()
assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
del x
qs = normalized_queries(1100, dim, 3)
h = C.c_void_p()
def loop(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        q = qs[i % 1100]
        assert L.vt_flat_search(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), dim, 10, C.byref(h)) == 0
        L.vt_hits_free(h)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
loop(200)
for rep in range(3):
    nifs.flat_set_profiling(ref, False); a = loop(1000)
    nifs.flat_set_profiling(ref, True); b = loop(1000)
    p = nifs.flat_get_profile(ref, reset=True)
    print("no-events %.4f ms   with-events %.4f ms   kernel %.4f ms" % (a, b, p["scan_ms"] / max(1, p["scan_launches"])))
