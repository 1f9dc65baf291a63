import sys, ctypes as C, numpy as np, time, os
sys.path.insert(0, '/root/repo')
import torch
from vettore_amd import nifs, _lib
from bench import build_shard, doc_ids
L = _lib.load()
for rows in (500000, 2000000, 4000000):
    dim = 768
    x = build_shard(torch, torch.device("cuda", 0), rows, dim, 99)
    ref = nifs._flat_new(2)
    assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
    q = np.random.default_rng(0).uniform(-1, 1, dim).astype(np.float32); q /= np.linalg.norm(q)
    qp = q.ctypes.data_as(C.POINTER(C.c_float)); h = C.c_void_p()
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        assert L.vt_flat_search(ref.handle, qp, dim, 10, C.byref(h)) == 0
        ts.append(round((time.perf_counter() - t0) * 1e3, 3))
        L.vt_hits_free(h)
    print(rows, ts, flush=True)
    del ref, x
    torch.cuda.empty_cache()
