#!/usr/bin/env python3
"""Soak test for float hamming / jaccard on corpora with a non-zero-bit column: a base of
17 000 sparse rows, then random insert / upsert / delete / search / batch interleavings under fresh
seeds for SECONDS seconds, every answer compared bit for bit with the oracle index that saw the same
operations (ids arrive in any order: lazy ranks, boundary ties and the column's per-row patches).
    SECONDS=120 python tools/soak_pattern.py
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import oracle
from vettore_amd import nifs
oracle.build()


def bits(h): return [(x[0], np.float32(x[1]).tobytes()) for x in h]


def sparse(rng, shape, density):
    return (rng.uniform(-1, 1, shape) * (rng.uniform(0, 1, shape) < density)).astype(np.float32)


def run(seed, m, steps=150):
    rng = np.random.default_rng(seed)
    d = int(rng.choice([24, 64, 100, 130]))
    density = float(rng.uniform(0.1, 0.7))
    n0 = 17_000
    ref = nifs._flat_new(m)
    o = oracle.FlatIndex(m)
    base = [("base-%05d" % i, v) for i, v in enumerate(sparse(rng, (n0, d), density))]
    assert nifs.flat_insert_many(ref, base)[0] == "ok"
    o.insert_many(base)
    nifs.flat_set_profiling(ref, True)
    for step in range(steps):
        op = rng.integers(0, 10)
        if op < 5:
            cnt = int(rng.integers(1, 30))
            items = [("%s-%d" % (rng.choice(["aa", "base", "zz"]), rng.integers(0, 900)), sparse(rng, d, density)) for _ in range(cnt)]
            assert nifs.flat_insert_many(ref, items)[0] == "ok"
            o.insert_many(items)
        elif op < 7:
            victim = "base-%05d" % rng.integers(0, n0)
            nifs.flat_delete(ref, victim); o.delete(victim)
        elif op < 8:
            victim = "zz-%d" % rng.integers(0, 900)
            nifs.flat_delete(ref, victim); o.delete(victim)
        assert len(ref) == len(o)
        q = sparse(rng, d, density) if rng.integers(0, 8) else np.zeros(d, np.float32)
        k = int(rng.choice([1, 3, 10, 40, 64, 65, 300]))
        if rng.integers(0, 3) == 0:
            qs = np.stack([q] + [sparse(rng, d, density) for _ in range(int(rng.integers(1, 12)))])
            got = nifs.flat_search_batch(ref, qs, k)[1]
            ok = all(bits(got[i]) == bits(o.search(qs[i], k)) for i in range(len(qs)))
        else:
            ok = bits(nifs.flat_search(ref, q, k)[1]) == bits(o.search(q, k))
        if not ok:
            print("MISMATCH seed", seed, "metric", m, "step", step, "k", k, "d", d, "n", len(o), flush=True)
            return False
    prof = nifs.flat_get_profile(ref, reset=True)
    assert prof["hamming_launches"] > 0, prof   # (the column was in use)
    return True


t0 = time.time()
bad = runs = 0
budget = float(os.environ.get("SECONDS", 120))
seed = int(os.environ.get("SEED", int(time.time()) % 100000))
while time.time() - t0 < budget:
    for m in (7, 8):
        runs += 1
        if not run(seed, m):
            bad += 1
        seed += 1
print("runs", runs, "first seed", seed - runs, "mismatches", bad)
sys.exit(1 if bad else 0)
