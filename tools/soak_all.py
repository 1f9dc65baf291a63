#!/usr/bin/env python3
"""Soak test over every search entry point: random insert / upsert / delete steps on a small index
(plain or multi-shard), and after every step a search -- plain (limits 1..300), batched, quantized,
funnel (one or two stages), hybrid -- compared bit for bit with the oracle (an oracle index for the flat searches, the
reference's compositions binary_top_k -> vector_top_k / prefix vector_top_k -> rerank for the staged
ones).  Fresh seeds until SECONDS are over.
    SECONDS=200 SHARDS=3 METRICS=0,2 DIM=64 python tools/soak_all.py
    VT_SLAB_CHUNK_MB=1 IDS=9000 BATCH=600 DIM=100 ...   (rows cross the mapped slab's chunk borders)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import oracle  # noqa: E402
from vettore_amd import nifs  # noqa: E402

oracle.build()


def bits(h):
    return [(x[0], np.float32(x[1]).tobytes()) for x in h]


def ok(res):
    assert res[0] == "ok", res
    return res[1]


IDS = int(os.environ.get("IDS", 700))        # id space: the index holds up to this many rows
BATCH = int(os.environ.get("BATCH", 40))     # rows per insert_many, at most


def run(seed, metric, d, shards, steps=300):
    rng = np.random.default_rng(seed)
    ref = nifs.flat_new_sharded(metric, [0] * shards) if shards else nifs._flat_new(metric)
    o = oracle.FlatIndex(metric)
    mirror = {}

    def vec():
        v = rng.uniform(-1, 1, d).astype(np.float32)
        if rng.integers(0, 4) == 0:
            v = np.round(v * 2) / 2            # coarse coordinates: exact ties
        return oracle.normalize_l2(v) if metric == 2 else v

    for step in range(steps):
        op = rng.integers(0, 10)
        if op < 5 or not mirror:
            items = [(b"id-%d" % rng.integers(0, IDS), vec()) for _ in range(int(rng.integers(1, BATCH)))]
            ok(nifs.flat_insert_many(ref, items))
            o.insert_many(items)
            mirror.update(items)
        elif op < 8:
            victim = list(mirror)[int(rng.integers(0, len(mirror)))]
            ok(nifs.flat_delete(ref, victim))
            o.delete(victim)
            del mirror[victim]
        assert len(ref) == len(o) == len(mirror)
        if not mirror:
            continue
        q = vec()
        what = step % 13
        try:
            if what in (3, 9):
                nq = int(rng.integers(2, 10))
                qs = np.stack([vec() for _ in range(nq)])
                k = int(rng.integers(1, 33))
                got = ok(nifs.flat_search_batch(ref, qs, k))
                for i in range(nq):
                    assert bits(got[i]) == bits(o.search(qs[i], k)), ("batch", i, k)
            elif what == 5:
                cand = int(rng.integers(1, 60)); k = int(rng.integers(1, 20))
                got = ok(nifs.flat_quantized_search(ref, q, cand, k))
                rows = list(mirror.items())
                c = oracle.binary_top_k([(i, oracle.compress_sign_bits(v)) for i, v in rows], oracle.compress_sign_bits(q), d, cand)
                want = oracle.vector_top_k([(i, mirror[i]) for i, _ in c], q, metric, d, k)
                assert bits(got) == bits(want), ("quantized", cand, k)
            elif what == 11:
                cand = int(rng.integers(1, 60)); k = int(rng.integers(1, 20))
                pres = sorted(int(p) for p in rng.integers(1, d + 1, size=int(rng.integers(1, 3))))
                got = ok(nifs.flat_funnel_search(ref, q, pres, cand, k))
                c = list(mirror.items())
                for pre in pres:
                    c = [(i, mirror[i]) for i, _ in oracle.vector_top_k(c, q, metric, pre, cand)]
                want = oracle.vector_top_k(c, q, metric, d, k)
                assert bits(got) == bits(want), ("funnel", pres, cand, k)
            elif what == 1:
                # hybrid_search, rerank: :exact (collection.ex:325-345): the union of the generators' candidates
                cf, cq, cs = (int(v) for v in rng.integers(1, 40, size=3)); k = int(rng.integers(1, 20))
                pre = int(rng.integers(1, d + 1))
                got = ok(nifs.flat_hybrid_search(ref, q, [(nifs.GEN_FUNNEL, cf, [pre]), (nifs.GEN_QUANTIZED, cq, []),
                                                          (nifs.GEN_SEARCH, cs, [])], k))
                rows = list(mirror.items())
                union = {i for i, _ in oracle.vector_top_k(rows, q, metric, pre, cf)}
                union |= {i for i, _ in oracle.binary_top_k([(i, oracle.compress_sign_bits(v)) for i, v in rows],
                                                            oracle.compress_sign_bits(q), d, cq)}
                union |= {i for i, _ in o.search(q, cs)}
                want = oracle.vector_top_k([(i, mirror[i]) for i in union], q, metric, d, k)
                assert bits(got) == bits(want), ("hybrid", pre, cf, cq, cs, k)
            else:
                k = int(rng.integers(1, 30)) if what != 7 else int(rng.integers(250, 320))
                assert bits(ok(nifs.flat_search(ref, q, k))) == bits(o.search(q, k)), ("search", k)
        except AssertionError as e:
            print("MISMATCH seed", seed, "metric", metric, "d", d, "shards", shards, "step", step, "n", len(mirror), e.args, flush=True)
            return False
    return True


def main():
    budget = float(os.environ.get("SECONDS", 120))
    shards = int(os.environ.get("SHARDS", 0))
    metrics = [int(m) for m in os.environ.get("METRICS", "0,2,3,5").split(",")]
    dims = [int(v) for v in os.environ.get("DIM", "16,64,100").split(",")]
    t0 = time.time()
    seed, runs, bad = int(os.environ.get("SEED", 1000)), 0, 0
    while time.time() - t0 < budget:
        for m in metrics:
            for d in dims:
                runs += 1
                bad += 0 if run(seed, m, d, shards) else 1
                seed += 1
    print("runs", runs, "mismatches", bad, "shards", shards)


if __name__ == "__main__":
    main()
