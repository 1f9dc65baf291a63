#!/usr/bin/env python3
"""How large an index grows on one card when the rows arrive in batches (not as one bulk load):
appends device-resident batches of STEP rows x DIM until TARGET_GB of rows are resident or an
append fails, printing the slab's size after each.  With the mapped slab nothing is ever copied
and the slab is the rows plus at most one 1-GiB chunk; `VT_SLAB=malloc` shows the
allocate-and-copy form for comparison (it needs old + new slab side by side).
    TARGET_GB=230 STEP=2000000 python tools/capacity_probe.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vettore_amd import nifs  # noqa: E402
from bench import build_shard, doc_ids  # noqa: E402


def main():
    d = int(os.environ.get("DIM", 768))
    step = int(os.environ.get("STEP", 2_000_000))
    target = float(os.environ.get("TARGET_GB", 200)) * 1e9
    dev = torch.device("cuda", 0)
    ref = nifs._flat_new(2)
    x = build_shard(torch, dev, step, d, 1)          # the same batch every time under fresh ids: the probe is about memory
    first = x[:4].cpu().numpy()
    total, t0 = 0, time.perf_counter()
    worst_append = 0.0
    while total * d * 4 < target:
        ids = doc_ids(total, step)
        t1 = time.perf_counter()
        res = nifs.flat_load_device_matrix(ref, ids, x.data_ptr(), step, d)
        dt = time.perf_counter() - t1
        if res != ("ok", ()):
            print(json.dumps({"append_failed_at_rows": total, "error": res}), flush=True)
            break
        total += step
        worst_append = max(worst_append, dt)
        cap, nbytes, chunks = nifs.flat_shard_memory(ref)
        free, tot = torch.cuda.mem_get_info(0)
        print(json.dumps({"rows": total, "row_GB": round(total * d * 4 / 1e9, 1), "slab_GB": round(nbytes / 1e9, 1),
                          "chunks": chunks, "append_ms": round(dt * 1e3, 1), "device_free_GB": round(free / 1e9, 1)}), flush=True)
    st, hits = nifs.flat_search(ref, first[0], 3)
    t1 = time.perf_counter()
    for _ in range(5):
        nifs.flat_search(ref, first[1], 10)
    ms = (time.perf_counter() - t1) / 5 * 1e3
    print(json.dumps({"final_rows": len(ref), "row_GB": round(len(ref) * d * 4 / 1e9, 1), "search_ms": round(ms, 2),
                      "scan_GBps": round(len(ref) * d * 4 / ms / 1e6), "self_hit_raw": hits[0][1] if st == "ok" else None,
                      "first_hits": [h[0].decode() for h in hits] if st == "ok" else [st, str(hits)],
                      "worst_append_ms": round(worst_append * 1e3, 1), "seconds": round(time.perf_counter() - t0, 1),
                      "slab": os.environ.get("VT_SLAB", "mapped")}), flush=True)


if __name__ == "__main__":
    main()
