"""flat_search under float hamming / jaccard: the K4 pass over the non-zero-bit column against
the K1 scan of the rows.  One JSON line per metric.  (The K1 leg was a second process under VT_NO_PATTERN_BITS=1, a switch
that left the library in r06 with the other A/B switches: its numbers are profiles/r04_pattern_* / r05_pattern_*; corpora
below 16 384 rows still take K1.)
    ROWS=10000000 D=768 python3 tools/pattern_probe.py
"""
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (first: the HIP runtime comes up through torch's copy)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vettore_amd import nifs  # noqa: E402

rows, d = int(os.environ.get("ROWS", "10000000")), int(os.environ.get("D", "768"))
rng = np.random.default_rng(5)
only = [int(v) for v in os.environ.get("METRICS", "7,8").split(",")]   # METRICS=7: float hamming alone (the pass bench.py's side leg prices)
for metric, name in ((7, "hamming"), (8, "jaccard")):
    if metric not in only:
        continue
    ref = nifs._flat_new(metric)
    chunk = 500_000
    for at in range(0, rows, chunk):
        m = min(chunk, rows - at)
        x = rng.random((m, d), dtype=np.float32) - np.float32(0.5)
        x *= rng.random((m, d), dtype=np.float32) < 0.5
        ids = [b"%09d" % (at + i) for i in range(m)]
        r = nifs.flat_load_matrix(ref, ids, x)
        assert r == "ok" or r[0] == "ok", r
    qs = (rng.uniform(-1, 1, (64, d)) * (rng.uniform(0, 1, (64, d)) < 0.5)).astype(np.float32)
    nifs.flat_set_profiling(ref, True)
    for q in qs[:4]:
        nifs.flat_search(ref, q, 10)
    nifs.flat_get_profile(ref, reset=True)
    t0 = time.perf_counter()
    for q in qs:
        hits = nifs.flat_search(ref, q, 10)
    dt = (time.perf_counter() - t0) / len(qs)
    prof = nifs.flat_get_profile(ref, reset=True)
    nifs.flat_search_batch(ref, qs[:16], 10)
    t0 = time.perf_counter()
    nifs.flat_search_batch(ref, qs[:16], 10)
    dt16 = time.perf_counter() - t0
    nifs.flat_search_batch(ref, qs, 10)
    t0 = time.perf_counter()
    nifs.flat_search_batch(ref, qs, 10)
    dt64 = time.perf_counter() - t0
    print(json.dumps({"metric": name, "rows": rows, "d": d, "pattern_bits": True,
                      "ms_per_search": round(dt * 1e3, 4), "ms_per_batch_of_16": round(dt16 * 1e3, 3), "ms_per_batch_of_64": round(dt64 * 1e3, 3),
                      "hamming_launches": prof["hamming_launches"], "hamming_ms_per_launch":
                      round(prof["hamming_ms"] / max(1, prof["hamming_launches"]), 4),
                      "scan_launches": prof["scan_launches"],
                      "scan_ms_per_launch": round(prof["scan_ms"] / max(1, prof["scan_launches"]), 4),
                      "first_hit": [hits[1][0][0].decode(), float(hits[1][0][1])]}), flush=True)
    del ref
