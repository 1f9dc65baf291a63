#!/usr/bin/env python3
"""One batched search workload for profiling (rocprofv3 -- python3 tools/batch_probe.py)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vettore_amd import nifs, _lib  # noqa: E402
from bench import build_shard, doc_ids  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dim = 768
dev = torch.device("cuda", 0)
x = build_shard(torch, dev, rows, dim, 1234)
ref = nifs._flat_new(3)
assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
del x
qs = np.random.default_rng(1).uniform(-1, 1, size=(nq, dim)).astype(np.float32)
L = _lib.load()
outs = (C.c_void_p * nq)()
for _ in range(reps):
    assert L.vt_flat_search_batch(ref.handle, qs.ctypes.data_as(C.POINTER(C.c_float)), nq, dim, 10, outs) == 0
    for i in range(nq):
        L.vt_hits_free(C.c_void_p(outs[i]))
print("done")
