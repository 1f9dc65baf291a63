import sys, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from vettore_amd import nifs, _lib
from bench import build_shard, doc_ids
L=_lib.load()
rows, dim = 200000, 768
x = build_shard(torch, torch.device("cuda",0), rows, dim, 7)
ref = nifs._flat_new(2)
assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
v = np.random.default_rng(1).uniform(-1,1,(40,dim)).astype(np.float32); v/=np.linalg.norm(v,axis=1,keepdims=True)
for i in range(20):
    idb=b"new-%07d"%i
    assert L.vt_flat_insert(ref.handle, idb, len(idb), v[i].ctypes.data_as(C.POINTER(C.c_float)), dim)==0
nifs.debug_set("trace_ingest", 2)
for i in range(20,24):
    idb=b"new-%07d"%i
    assert L.vt_flat_insert(ref.handle, idb, len(idb), v[i].ctypes.data_as(C.POINTER(C.c_float)), dim)==0
print("upserts", file=sys.stderr)
for i in range(20,24):
    idb=b"new-%07d"%i
    assert L.vt_flat_insert(ref.handle, idb, len(idb), v[i].ctypes.data_as(C.POINTER(C.c_float)), dim)==0
