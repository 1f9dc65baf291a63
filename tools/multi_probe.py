#!/usr/bin/env python3
"""K1m (several queries per sweep): time per batch and per sweep against the single scan, for a
GEMM-less metric and (with the matrix cores switched off) a dot-family one.  Diagnostic only."""
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
    rows = int(os.environ.get("ROWS", 2_000_000))
    dim = int(os.environ.get("DIM", 768))
    nifs.debug_set("batch_no_mfma", 1)
    # (since r05 wide rows of a corpus of 2 GB and more go to K1p: K1m itself is what DIM=384, or ROWS below 650 000 at 768, shows)
    rng = np.random.default_rng(0)
    for metric in (int(m) for m in os.environ.get("METRICS", "5,2,0,7").split(",")):
        x = build_shard(torch, torch.device("cuda", 0), rows, dim, 99)
        ref = nifs._flat_new(metric)
        assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
        del x
        nifs.flat_set_profiling(ref, True)
        out = {"metric": nifs.METRICS[metric], "rows": rows, "dim": dim}
        for nq in (int(v) for v in os.environ.get("NQS", "1,2,4,8,16,32").split(",")):
            qs = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
            qsp = qs.ctypes.data_as(C.POINTER(C.c_float))
            outs = (C.c_void_p * nq)()

            def call():
                if nq == 1:
                    assert L.vt_flat_search(ref.handle, qsp, dim, 10, outs) == 0
                else:
                    assert L.vt_flat_search_batch(ref.handle, qsp, nq, dim, 10, outs) == 0
                for i in range(nq):
                    L.vt_hits_free(C.c_void_p(outs[i]))

            call()
            nifs.flat_get_profile(ref, reset=True)
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            dt = (time.perf_counter() - t0) / reps
            p = nifs.flat_get_profile(ref, reset=True)
            sweep_ms = p["scan_ms"] / max(1, p["scan_launches"])
            out["nq%d" % nq] = {"call_us": round(dt * 1e6, 1), "sweeps": p["scan_launches"] // reps,
                                "sweep_ms": round(sweep_ms, 4), "GBps": round(rows * dim * 4 / sweep_ms / 1e6, 0)}
        print(json.dumps(out), flush=True)
        del ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
