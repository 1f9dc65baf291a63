#!/usr/bin/env python3
"""Soak test: random insert / upsert / delete / search interleavings (the scenario of
tests/test_gpu_parity.py::test_mutations_follow_the_oracle) under fresh seeds for SECONDS seconds,
every search compared bit for bit with the oracle index that saw the same operations.  This is
what caught the equal-keys bug of the lazy-rank mode (1 wrong answer in ~1e6 searches).
    SECONDS=240 python tools/soak_mutations.py
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import oracle
from vettore_amd import nifs
oracle.build()

def bits(h): return [(x[0], np.float32(x[1]).tobytes()) for x in h]

def run(seed, m, steps=400, d=16):
    rng = np.random.default_rng(seed)
    ref = nifs._flat_new(m)
    nifs.flat_set_reduce_order(ref, 3)
    o = oracle.FlatIndex(m)
    live = []
    for step in range(steps):
        op = rng.integers(0, 10)
        if op < 5 or not live:
            cnt = int(rng.integers(1, 40))
            items = [("id-%d" % rng.integers(0, 600), rng.uniform(-1, 1, d).astype(np.float32)) for _ in range(cnt)]
            assert nifs.flat_insert_many(ref, items)[0] == "ok"
            o.insert_many(items)
            live = list({*live, *[i for i, _ in items]})
        elif op < 8:
            victim = live.pop(int(rng.integers(0, len(live))))
            nifs.flat_delete(ref, victim); o.delete(victim)
        else:
            nifs.flat_delete(ref, "missing-%d" % step); o.delete("missing-%d" % step)
        assert len(ref) == len(o)
        q = rng.uniform(-1, 1, d).astype(np.float32)
        k = int(rng.integers(1, 30))
        if len(o) == 0:
            continue
        got = nifs.flat_search(ref, q, k)[1]
        want = o.search(q, k)
        if bits(got) != bits(want):
            full_g = nifs.flat_search(ref, q, len(o))[1]
            full_o = o.search(q, len(o))
            gd = dict(full_g)
            first = next(i for i, (a, b) in enumerate(zip(got, want)) if bits([a]) != bits([b]))
            print("MISMATCH seed", seed, "metric", m, "step", step, "k", k, "n", len(o), "first", first, got[first], want[first],
                  "gpu raw of wanted id", gd.get(want[first][0]), "full equal", bits(full_g) == bits(full_o),
                  "again equal", bits(nifs.flat_search(ref, q, k)[1]) == bits(want), flush=True)
            return False
    return True

t0 = time.time()
bad = runs = 0
budget = float(os.environ.get("SECONDS", 150))
seed = 5
while time.time() - t0 < budget:
    for m in (0, 2, 3):
        runs += 1
        if not run(5 if runs % 2 else seed, m):
            bad += 1
    seed += 1
print("runs", runs, "mismatches", bad)
