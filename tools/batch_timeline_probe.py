import sys, ctypes as C, numpy as np, time
sys.path.insert(0, '/root/repo')
import torch
from vettore_amd import nifs, _lib
from bench import build_shard, doc_ids
L = _lib.load()
rows, dim = 130000, 768
x = build_shard(torch, torch.device("cuda", 0), rows, dim, 99)
ref = nifs._flat_new(2)
assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
qs = np.random.default_rng(0).uniform(-1, 1, (16, dim)).astype(np.float32)
qsp = qs.ctypes.data_as(C.POINTER(C.c_float))
outs = (C.c_void_p * 16)()
for it in range(6):
    t0 = time.perf_counter()
    assert L.vt_flat_search_batch(ref.handle, qsp, 16, dim, 10, outs) == 0
    t1 = time.perf_counter()
    for i in range(16): L.vt_hits_free(C.c_void_p(outs[i]))
    print("batch us", round((t1 - t0) * 1e6, 1), flush=True)
