#!/usr/bin/env python3
"""Soak of round 5's pipelined batch calls: vt_flat_search_batch with 2..6 groups of 256 queries (consecutive groups
alternate between two contexts, group g + 1 queued before group g is waited for; the K2s sample pass files group maxima)
on corpora of 17 000-40 000 rows under the five metrics that take the matrix cores, both nominations, after random
mutations, with a second thread running batches of its own on the same handle (it competes for the spare contexts) --
every list against the same query searched alone, a handful per batch against the oracle.
SECONDS / SEED / METRICS / SHARDS env.  Prints one line per run and a summary; diagnostic only (tests/ hold the fixed cases)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from vettore_amd import nifs  # noqa: E402

nifs.debug_set("force_batch_mfma", 1)   # (the cost model would send corpora this small to sweeps and single scans)


def bits(hits):
    return [(h[0], np.float32(h[1]).tobytes()) for h in hits]


def ok(res):
    assert res[0] == "ok", res
    return res[1]


COUNTS = {"batches": 0, "queries": 0, "groups": 0, "oracle_checks": 0, "fallbacks": 0, "passes": 0}


def run(seed, metric):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(17_000, 40_000))
    d = int(rng.choice([64, 128, 192, 256]))
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    if rng.integers(0, 3) == 0:
        x = np.round(x * 4) / 4                       # coarse coordinates: exact ties everywhere
    blk = int(rng.integers(2, 200))
    x[1000:1000 + blk] = x[1000]                      # identical rows: only the ids order them
    if metric == 2:
        x = np.stack([oracle.normalize_l2(r) for r in x])
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    shards = int(os.environ.get("SHARDS", "0"))
    # (SHARDS=n: a sharded handle with n shards on device 0 -- the groups pipeline inside every shard, and the calling
    # thread merges the shards' lists query by query while later groups still run)
    ref = nifs.flat_new_sharded(metric, [0] * shards) if shards else nifs._flat_new(metric)
    assert nifs.flat_set_batch_nominate(ref, 1 if rng.integers(0, 4) == 0 else 2) == "ok"   # (mostly bf16, the default)
    ok(nifs.flat_load_matrix(ref, ids, x))
    nifs.flat_set_profiling(ref, True)
    rows = dict(zip(ids, x))
    for _ in range(int(rng.integers(0, 5))):          # derived columns get patched, ranks go lazy and come back
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
    mat, packed = np.stack([v for _, v in items]), oracle.pack_ids([i for i, _ in items])

    def queries(count, r):
        qs = r.uniform(-1, 1, (count, d)).astype(np.float32)
        for j in range(0, count, 97):
            qs[j] = x[int(r.integers(0, n))]
        return np.stack([oracle.normalize_l2(q) for q in qs]) if metric == 2 else qs

    # a second caller on the same handle: batches of two groups, checked against single searches
    other = {"bad": None, "stop": False}

    def second_caller():
        r2 = np.random.default_rng(seed + 77)
        while not other["stop"] and other["bad"] is None:
            qs2 = queries(int(r2.integers(257, 600)), r2)
            k2 = int(r2.integers(1, 20))
            got2 = ok(nifs.flat_search_batch(ref, qs2, k2))
            for i in (0, 255, 256, len(qs2) - 1):
                if bits(got2[i]) != bits(ok(nifs.flat_search(ref, qs2[i], k2))):
                    other["bad"] = ("second caller", k2, i)

    th = threading.Thread(target=second_caller)
    th.start()
    good = True
    try:
        for step in range(3):
            nq = int(rng.integers(257, 1500))
            k = int(rng.integers(1, 65))
            qs = queries(nq, rng)
            got = ok(nifs.flat_search_batch(ref, qs, k))
            COUNTS["batches"] += 1
            COUNTS["queries"] += nq
            COUNTS["groups"] += (nq + 255) // 256
            check = sorted(set(range(0, nq, 11)) | set(range(255, nq, 256)) | set(range(0, nq, 256)) | {nq - 1})
            for i in check:
                assert bits(got[i]) == bits(ok(nifs.flat_search(ref, qs[i], k))), ("batch vs single", nq, k, i)
            for i in check[::9]:
                COUNTS["oracle_checks"] += 1
                assert bits(got[i]) == bits(oracle.matrix_search(metric, mat, packed, qs[i], k)), ("batch vs oracle", nq, k, i)
    except AssertionError as e:
        print("MISMATCH seed", seed, "metric", metric, "n", n, "d", d, e.args, flush=True)
        good = False
    other["stop"] = True
    th.join()
    if other["bad"] is not None:
        print("MISMATCH seed", seed, "metric", metric, "n", n, "d", d, other["bad"], flush=True)
        good = False
    p = nifs.flat_get_profile(ref)
    COUNTS["fallbacks"] += p["batch_fallbacks"]
    COUNTS["passes"] += p["nominate_launches"] + p["batch_launches"]
    return good


def main():
    budget = float(os.environ.get("SECONDS", 120))
    metrics = [int(m) for m in os.environ.get("METRICS", "0,1,2,3,4").split(",")]
    seed, runs, bad, t0 = int(os.environ.get("SEED", int(time.time()) % 100000)), 0, 0, time.time()
    first = seed
    while time.time() - t0 < budget:
        for metric in metrics:
            if not run(seed, metric):
                bad += 1
            runs += 1
            seed += 1
    print("runs", runs, "first seed", first, "mismatches", bad, "seconds", round(time.time() - t0, 1), COUNTS)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
