#!/usr/bin/env python3
"""Soak test of the handle's shared / exclusive locking: READERS threads run plain, large-limit,
batched, quantized and funnel searches on ONE handle (plain or multi-shard) while a writer thread
inserts, upserts and deletes rows that can never reach a result list (far from every query; their
ids sort both before and after the corpus ids, so the lazy-rank mode, its escalations and the
derived columns' upkeep all run under load).  Every answer must equal the one computed up front.
    SECONDS=180 SHARDS=0 python tools/soak_concurrent.py
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from vettore_amd import nifs  # noqa: E402


def bits(h):
    return [(x[0], np.float32(x[1]).tobytes()) for x in h]


def ok(res):
    assert res[0] == "ok", res
    return res[1]


def main():
    budget = float(os.environ.get("SECONDS", 120))
    shards = int(os.environ.get("SHARDS", 0))
    readers = int(os.environ.get("READERS", 8))
    metric = int(os.environ.get("METRIC", 0))
    n, d = int(os.environ.get("N", 30000)), 64
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    x[n // 2:n // 2 + 24] = x[n // 2]                      # a block of identical rows: ties by id
    if metric == 2:
        x /= np.linalg.norm(x.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    ref = nifs.flat_new_sharded(metric, [0] * shards) if shards else nifs._flat_new(metric)
    ok(nifs.flat_load_matrix(ref, ids, x))
    qs = [x[n // 2]] + [rng.uniform(-1, 1, d).astype(np.float32) for _ in range(7)]
    if metric == 2:
        qs = [(q / np.linalg.norm(q)).astype(np.float32) for q in qs]
    want = []
    for q in qs:
        want.append({
            "s": bits(ok(nifs.flat_search(ref, q, 10))),
            "big": bits(ok(nifs.flat_search(ref, q, 300))),
            "q": bits(ok(nifs.flat_quantized_search(ref, q, 100, 10))),
            "f": bits(ok(nifs.flat_funnel_search(ref, q, [32], 100, 10))),
        })
    far = (rng.uniform(-1, 1, (500, d)) * 0.01 + (40.0 if metric != 2 else 0.0)).astype(np.float32)
    if metric == 2:   # cosine: "far" = opposite of every query direction is not possible; use a fixed direction never close
        far = -np.abs(far) - 1.0
        far /= np.linalg.norm(far.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    errors, stop = [], threading.Event()
    counts = [0] * readers

    def reader(t):
        r = np.random.default_rng(100 + t)
        try:
            while not stop.is_set():
                j = int(r.integers(0, len(qs)))
                mode = int(r.integers(0, 6))
                if mode == 0:
                    good = bits(ok(nifs.flat_search(ref, qs[j], 10))) == want[j]["s"]
                elif mode == 1:
                    good = bits(ok(nifs.flat_search(ref, qs[j], 300))) == want[j]["big"]
                elif mode == 2:
                    got = ok(nifs.flat_search_batch(ref, np.stack([qs[j], qs[(j + 1) % len(qs)], qs[j]]), 10))
                    good = bits(got[0]) == want[j]["s"] and bits(got[2]) == want[j]["s"]
                elif mode == 3:
                    good = bits(ok(nifs.flat_quantized_search(ref, qs[j], 100, 10))) == want[j]["q"]
                elif mode == 4:
                    good = bits(ok(nifs.flat_funnel_search(ref, qs[j], [32], 100, 10))) == want[j]["f"]
                else:
                    good = bits(ok(nifs.flat_search(ref, qs[j], 1))) == want[j]["s"][:1]
                counts[t] += 1
                if not good:
                    errors.append(("reader", t, mode, j, counts[t]))
                    stop.set()
        except Exception as e:  # noqa: BLE001
            errors.append(("reader", t, repr(e)))
            stop.set()

    def writer():
        r = np.random.default_rng(9)
        i = 0
        try:
            while not stop.is_set():
                kind = int(r.integers(0, 4))
                j = int(r.integers(0, 500))
                key = (b"aa-far-%d" if j % 2 else b"zz-far-%d") % j
                if kind < 2:
                    ok(nifs.flat_insert(ref, key, far[j]))
                elif kind == 2:
                    ok(nifs.flat_delete(ref, key))
                else:
                    ok(nifs.flat_insert_many(ref, [((b"mm-far-%d" % ((j + t) % 300)), far[(j * 3 + t) % 500]) for t in range(5)]))
                i += 1
        except Exception as e:  # noqa: BLE001
            errors.append(("writer", repr(e)))
            stop.set()
        counts.append(i)

    ths = [threading.Thread(target=reader, args=(t,)) for t in range(readers)] + [threading.Thread(target=writer)]
    for th in ths:
        th.start()
    time.sleep(budget)
    stop.set()
    for th in ths:
        th.join()
    print("shards", shards, "metric", metric, "reads", sum(counts[:readers]), "writes", counts[-1], "errors", errors[:3])


if __name__ == "__main__":
    main()
