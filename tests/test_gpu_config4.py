"""BASELINE.json configs[3] at its own size on ONE MI355X: `index: :flat, metric: :l2, d=768,
N=40M` (122.9 GB of rows -- the card's 288 GB holds it) as one handle of eight shards, all on
device 0: the same hash routing, worker threads, exchange and (rank key, id bytes) merge an
8-GPU node runs, only the devices coincide.  The expected hits are exact: while a shard's rows
are still in a torch tensor, a torch pass picks each query's 64 nearest rows of that shard (any
summation order does for a candidate set that generous); the oracle scores the 8 x 64
candidates in the reference's arithmetic and orders them by (rank, id bytes)."""
import time

import numpy as np
import pytest

import support

pytestmark = pytest.mark.gpu

N, D, SHARDS, CAND = 40_000_000, 768, 8, 64


def test_config4_forty_million_rows_eight_shards(oracle_mod):
    import torch
    from vettore_amd import nifs
    from bench import build_shard, doc_ids
    free, _total = torch.cuda.mem_get_info(0)
    if free < 160e9:
        pytest.skip("needs 160 GB of free HBM")
    dev = torch.device("cuda", 0)
    ref = nifs.flat_new_sharded(nifs.METRIC_CODE["l2"], [0] * SHARDS)
    route = nifs.flat_route_ids(ref, doc_ids(0, N))
    all_idx = np.arange(1, N + 1, dtype=np.int64)
    rng = np.random.default_rng(44)
    queries = [oracle_mod.normalize_l2(rng.uniform(-1, 1, D).astype(np.float32)) for _ in range(3)]
    planted = {}                                     # query number -> doc number whose row IS the query
    cands = [[] for _ in range(len(queries) + 2)]    # per query: (id bytes, f32 row)
    t0 = time.perf_counter()
    for s in range(SHARDS):
        idx = all_idx[route == s]
        x = build_shard(torch, dev, len(idx), D, 20260721 + s)
        if s in (2, 5):                              # two more queries: a stored row each (distance 0, ties with its duplicates)
            j = len(idx) // 3
            planted[len(queries)] = int(idx[j])
            queries.append(x[j].cpu().numpy())
        assert nifs.flat_load_device_matrix(ref, doc_ids(0, 0, idx), x.data_ptr(), len(idx), D) == ("ok", ())
        for qi, q in enumerate(queries):
            qd = torch.from_numpy(q).to(dev)
            near = torch.empty(len(idx), device=dev)
            step = 1 << 21
            for lo in range(0, len(idx), step):
                near[lo:lo + step] = (x[lo:lo + step] - qd).square_().sum(1)
            rows = torch.topk(near, CAND, largest=False).indices
            host = x[rows].cpu().numpy()
            for r, v in zip(rows.tolist(), host):
                cands[qi].append((b"doc-%d" % int(idx[r]), v))
            del near
        del x
        torch.cuda.empty_cache()
    # queries planted in a later shard were not offered to the earlier shards' candidate passes:
    # the first three queries are checked in full, the planted ones through their own properties
    load_s = time.perf_counter() - t0
    assert len(ref) == N and sum(nifs.flat_shard_lens(ref)) == N
    assert min(nifs.flat_shard_lens(ref)) > N // SHARDS * 0.99

    for qi in range(3):
        q = queries[qi]
        want = support.full_sort(cands[qi], lambda v: oracle_mod.compute(0, q, v),
                                 lambda raw: oracle_mod.rank_value(0, raw), 10)
        st, hits = nifs.flat_search(ref, q, 10)
        assert st == "ok"
        assert [(h[0], np.float32(h[1]).tobytes()) for h in hits] == \
               [(h[0], np.float32(h[1]).tobytes()) for h in want], qi
    for qi, doc in planted.items():
        st, hits = nifs.flat_search(ref, queries[qi], 10)
        assert st == "ok" and hits[0][1] == 0.0
        zero = [h[0] for h in hits if h[1] == 0.0]   # the row and its verbatim duplicates, by id bytes
        assert b"doc-%d" % doc in zero and zero == sorted(zero)
        keys = [(support.total_key(np.float32(h[1])), h[0]) for h in hits]
        assert keys == sorted(keys)
    # batched == single on the multi-shard handle at this size
    st, batch = nifs.flat_search_batch(ref, np.stack(queries), 10)
    assert st == "ok"
    for i, q in enumerate(queries):
        assert batch[i] == nifs.flat_search(ref, q, 10)[1]
    # what one query costs here: eight 15.4 GB scans sharing one GPU, then the merge
    t0 = time.perf_counter()
    reps = 20
    for i in range(reps):
        nifs.flat_search(ref, queries[i % 3], 10)
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("config4 on one GPU: %d rows in %d shards, load %.0f s, %.2f ms per query (%.0f GB/s over %.1f GB)" % (
        N, SHARDS, load_s, ms, N * D * 4 / ms / 1e6, N * D * 4 / 1e9))
    del ref
