#!/usr/bin/env python3
"""Creates, fills, queries (every search flavour) and frees indexes in a loop; device and host
memory must come back.  Diagnostic only."""
import gc
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import psutil  # noqa: E402
from vettore_amd import nifs  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    n, d = 20000, 128
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    ids = [b"doc-%d" % i for i in range(n)]
    proc = psutil.Process()
    for it in range(121):
        ref = nifs._flat_new(it % 9)
        assert nifs.flat_load_matrix(ref, ids, x) == ("ok", ())
        q = x[it]
        nifs.flat_search(ref, q, 10)
        nifs.flat_search(ref, q, 300)
        nifs.flat_search_batch(ref, x[:16], 5)
        if it % 9 in (0, 2, 3):
            nifs.flat_quantized_search(ref, q, 100, 10)
            nifs.flat_funnel_search(ref, q, [32, 64], 100, 10)
        nifs.flat_insert(ref, b"zz", q)
        nifs.flat_delete(ref, b"doc-5")
        nifs.flat_search(ref, q, 10)
        del ref
        gc.collect()
        if it % 40 == 0:
            free, total = torch.cuda.mem_get_info()
            print(json.dumps({"iter": it, "gpu_used_MB": round((total - free) / 1e6, 1),
                              "host_rss_MB": round(proc.memory_info().rss / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
