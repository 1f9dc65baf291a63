#!/usr/bin/env python3
"""Is the slow first second of scanning (bench.py --release-wait) a one-time effect after a load, or the card's
state after any idle stretch?  Scans the 10 M x 768 corpus in bursts separated by idle pauses of growing length and prints
the mean time of each burst's searches in windows of 50.  Diagnostic only."""
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
from bench import build_shard, doc_ids, normalized_queries  # noqa: E402

L = _lib.load()


def main():
    rows, dim = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("DIM", 768))
    x = build_shard(torch, torch.device("cuda", 0), rows, dim, 7)
    ref = nifs._flat_new(2)
    assert nifs.flat_load_device_matrix(ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
    # what could the slow first second be?  KEEP_SOURCE=1: the 30-GB source tensor is not returned to the driver;
    # SLEEP_AFTER_LOAD=<s>: the card idles between the load and the first scan (a background activity would be over by then)
    if not os.environ.get("KEEP_SOURCE"):
        del x
        torch.cuda.empty_cache()
    time.sleep(float(os.environ.get("SLEEP_AFTER_LOAD", "0")))
    qs = normalized_queries(64, dim, 5)
    hp = C.c_void_p()

    def burst(n):
        t = np.empty(n)
        for i in range(n):
            q = qs[i % 64]
            t0 = time.perf_counter()
            assert L.vt_flat_search(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), dim, 10, C.byref(hp)) == 0
            t[i] = time.perf_counter() - t0
            L.vt_hits_free(hp)
        w = int(os.environ.get("WINDOW", "50"))
        return [round(float(t[i:i + w].mean()) * 1e3, 4) for i in range(0, n, w)]

    out = {"rows": rows, "dim": dim, "bursts": []}
    L.vt_flat_search(ref.handle, qs[0].ctypes.data_as(C.POINTER(C.c_float)), dim, 10, C.byref(hp))   # (settles the id ranks)
    L.vt_hits_free(hp)
    out["keep_source"] = bool(os.environ.get("KEEP_SOURCE"))
    out["sleep_after_load_s"] = float(os.environ.get("SLEEP_AFTER_LOAD", "0"))
    for pause in [float(v) for v in os.environ.get("PAUSES", "0,0.1,0.5,1,2,5,10").split(",")]:
        time.sleep(pause)
        out["bursts"].append({"idle_before_s": pause, "ms_per_search_in_windows": burst(int(os.environ.get("BURST", "500")))})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
