#!/usr/bin/env python3
"""Soak of round 4's grouped paths on corpora large enough to take them (>= 16 384 rows): funnel batches under every
metric (K6bm / K1p / the bit column's prefix), plain batches as K1p sweeps (vt_debug_set "force_sweep_groups"), after random
mutations -- every answer against the same call made alone, two per batch against the oracle's composition.
SECONDS / SEED / METRICS env.  Prints one line per run and a summary; diagnostic only (tests/ hold the fixed cases)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from vettore_amd import nifs  # noqa: E402
nifs.debug_set("force_sweep_groups", 1)
nifs.debug_set("batch_no_mfma", 1)     # (the matrix cores have their own soak: tools/soak_all.py with larger batches)


def bits(hits):
    return [(h[0], np.float32(h[1]).tobytes()) for h in hits]


def ok(res):
    assert res[0] == "ok", res
    return res[1]


COUNTS = {"funnel_batches": 0, "plain_batches": 0, "queries": 0, "oracle_checks": 0, "grouped_queries": 0}


def run(seed, metric):
    rng = np.random.default_rng(seed)
    # (ROWS_MIN / ROWS_MAX: corpora of a few hundred thousand rows put the groups' sample on every n-th tile and take
    # their thresholds from tile maxima -- the form a 10 M-row corpus runs)
    n = int(rng.integers(int(os.environ.get("ROWS_MIN", 17_000)), int(os.environ.get("ROWS_MAX", 40_000))))
    d = int(rng.choice([24, 40, 64, 100, 130, 200, 260]))
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    if rng.integers(0, 2):
        x = np.round(x * 4) / 4                       # coarse coordinates: exact ties everywhere
    if metric in (7, 8):
        x[rng.uniform(size=x.shape) < 0.5] = 0.0
    blk = int(rng.integers(2, 300))
    x[1000:1000 + blk] = x[1000]                      # a block of identical rows: only the ids order them
    if metric == 2:
        x = np.stack([oracle.normalize_l2(r) for r in x])
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    ref = nifs._flat_new(metric)
    ok(nifs.flat_load_matrix(ref, ids, x))
    nifs.flat_set_profiling(ref, True)
    rows = dict(zip(ids, x))
    # a few mutations through the ordinary entry points (derived columns get patched, ranks go lazy)
    for _ in range(int(rng.integers(0, 6))):
        if rng.integers(0, 2):
            i = b"new-%d" % rng.integers(0, 10**6)
            v = x[int(rng.integers(0, n))].copy() if rng.integers(0, 2) else rng.uniform(-1, 1, d).astype(np.float32)
            if metric == 2:
                v = oracle.normalize_l2(v)
            ok(nifs.flat_insert(ref, i, v))
            rows[i] = v
        else:
            victim = ids[int(rng.integers(0, n))]
            if victim in rows:
                ok(nifs.flat_delete(ref, victim))
                del rows[victim]
    items = list(rows.items())
    mat, packed = np.stack([v for _, v in items]), oracle.pack_ids([i for i, _ in items])   # (flat_search's own oracle)

    def query():
        q = x[int(rng.integers(0, n))].copy() if rng.integers(0, 3) == 0 else rng.uniform(-1, 1, d).astype(np.float32)
        if metric in (7, 8):
            q[rng.uniform(size=d) < 0.5] = 0.0
        return oracle.normalize_l2(q) if metric == 2 else q

    for step in range(6):
        nq = int(rng.integers(2, 18)) if rng.integers(0, 4) else int(rng.integers(18, 60))   # (now and then five or more groups: they alternate between two contexts)
        qs = np.stack([query() for _ in range(nq)])
        COUNTS["queries"] += nq
        COUNTS["oracle_checks"] += 2
        try:
            if step % 2 == 0:
                stages = sorted(int(p) for p in rng.integers(1, d + 1, size=int(rng.integers(1, 3))))
                cand, k = int(rng.integers(1, 257)), int(rng.integers(1, 40))
                got = ok(nifs.flat_funnel_search_batch(ref, qs, stages, cand, k))
                COUNTS["funnel_batches"] += 1
                for i in range(nq):
                    assert bits(got[i]) == bits(ok(nifs.flat_funnel_search(ref, qs[i], stages, cand, k))), ("funnel batch vs single", stages, cand, k, i)
                for i in (0, nq - 1):
                    c = items
                    for pre in stages:
                        c = [(j, rows[j]) for j, _ in oracle.vector_top_k(c, qs[i], metric, pre, cand)]
                    assert bits(got[i]) == bits(oracle.vector_top_k(c, qs[i], metric, d, k)), ("funnel batch vs oracle", stages, cand, k, i)
            else:
                k = int(rng.integers(1, 257))
                got = ok(nifs.flat_search_batch(ref, qs, k))
                COUNTS["plain_batches"] += 1
                for i in range(nq):
                    assert bits(got[i]) == bits(ok(nifs.flat_search(ref, qs[i], k))), ("batch vs single", k, i)
                for i in (0, nq - 1):
                    assert bits(got[i]) == bits(oracle.matrix_search(metric, mat, packed, qs[i], k)), ("batch vs oracle", k, i)
        except AssertionError as e:
            print("MISMATCH seed", seed, "metric", metric, "n", n, "d", d, "step", step, e.args, flush=True)
            return False
    p = nifs.flat_get_profile(ref)
    COUNTS["grouped_queries"] += p["prefix_queries"] + p["sweep_queries"] + p["hamming_queries"]
    return True


def main():
    budget = float(os.environ.get("SECONDS", 120))
    metrics = [int(m) for m in os.environ.get("METRICS", "0,1,2,3,4,5,6,7,8").split(",")]
    seed, runs, bad, t0 = int(os.environ.get("SEED", 4000)), 0, 0, time.time()
    while time.time() - t0 < budget:
        for m in metrics:
            runs += 1
            bad += 0 if run(seed, m) else 1
            seed += 1
    print("runs", runs, "mismatches", bad, "seconds", round(time.time() - t0, 1), COUNTS)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
