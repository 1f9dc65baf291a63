#!/usr/bin/env python3
"""funnel_search under float hamming / jaccard: stage 1 from the prefix of the non-zero-bit column (K4 with a
prefix mask).  One JSON line per metric.  (Through r05 a second leg ran the same call with the column switched off --
VT_NO_PATTERN_BITS=1, K1 over the rows' prefixes: profiles/r04_pattern_funnel_probe.jsonl, r05_pattern_funnel_probe.jsonl;
the switch left the library in r06.)  ROWS / DIM / PREFIX env.  Diagnostic only."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def leg():
    import torch
    from vettore_amd import nifs, _lib
    from bench import build_shard, doc_ids
    L = _lib.load()
    rows, dim, prefix = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("DIM", 768)), int(os.environ.get("PREFIX", 128))
    dev = torch.device("cuda", 0)
    x = build_shard(torch, dev, rows, dim, 7)
    g = torch.Generator(device=dev)
    g.manual_seed(8)
    for s0 in range(0, rows, 1 << 20):
        e0 = min(rows, s0 + (1 << 20))
        x[s0:e0] *= (torch.rand((e0 - s0, dim), generator=g, device=dev) < 0.5)
    rng = np.random.default_rng(3)
    qs = (rng.uniform(-1, 1, (40, dim)) * (rng.uniform(0, 1, (40, dim)) < 0.5)).astype(np.float32)
    out = {"bits": True, "rows": rows, "dim": dim, "prefix": prefix}
    for metric in (7, 8):
        ref = nifs._flat_new(metric)
        assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
        st = (C.c_size_t * 1)(prefix)
        h = C.c_void_p()
        first = None
        times = []
        for i in range(40):
            t0 = time.perf_counter()
            assert L.vt_flat_funnel_search(ref.handle, qs[i].ctypes.data_as(C.POINTER(C.c_float)), dim, st, 1, 100, 10, C.byref(h)) == 0
            times.append(time.perf_counter() - t0)
            hits = nifs._take_hits(h)
            if i == 39:
                first = [(a.decode(), float(b)) for a, b in hits[:3]]
        out[nifs.METRICS[metric]] = {"call_ms": round(float(np.median(times[5:])) * 1e3, 4), "last_hits": first}
        del ref
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "leg":
        leg()
    else:
        subprocess.run([sys.executable, os.path.abspath(__file__), "leg"], env=dict(os.environ), check=True)
