import os, sys
sys.path.insert(0, os.getcwd())
import torch
import numpy as np
from vettore_amd import nifs
nifs.debug_set("batch_no_mfma", 1); nifs.debug_set("force_multi_scan", 1)
rng = np.random.default_rng(1)
for metric in (0, 3):
  for d in (64, 128, 320, 384, 576, 640, 832, 896):
    n = 4000
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    ids = [b"doc-%05d" % i for i in range(n)]
    ref = nifs._flat_new(metric)
    assert nifs.flat_load_matrix(ref, ids, x)[0] == "ok"
    qs = rng.uniform(-1, 1, (8, d)).astype(np.float32)
    got = nifs.flat_search_batch(ref, qs, 10)[1]
    bad = []
    for i in range(8):
        want = nifs.flat_search(ref, qs[i], 10)[1]
        if [(a, np.float32(b).tobytes()) for a, b in got[i]] != [(a, np.float32(b).tobytes()) for a, b in want]:
            # is the raw score of the same id different?
            wd = dict(want); diffs = [(a, b, wd.get(a)) for a, b in got[i] if wd.get(a) != b][:2]
            bad.append((i, diffs))
    print(metric, d, "mismatching queries:", [b[0] for b in bad], bad[:1])
