"""One resource, several shards (vt_flat_new_sharded, SURVEY.md 8e / 8b "device_mask") and
several readers on one handle (the reference's RwLock, nifs.rs:266-309) -- `-m gpu`.

The box has one GPU, so the shards of these tests all sit on device 0: every piece of the
multi-shard machinery runs (hash routing, one worker thread per shard, the per-shard
searches, the merge by (rank key, id bytes)) except the wire between devices.  The RCCL
exchange itself is exercised with a one-rank communicator (ncclCommInitAll over [0]).
Every result is compared bit for bit with the CPU oracle over ALL rows.
"""
import threading

import numpy as np
import pytest

import support
from support import load, run_steps
from test_gpu_parity import GpuError, GpuIndex, bits, make_corpus, nifs, unwrap  # noqa: F401  (nifs is a fixture)

pytestmark = pytest.mark.gpu


class ShardedIndex(GpuIndex):
    def __init__(self, nifs, metric_code, devices, order=3):
        self.n = nifs
        self.ref = nifs.flat_new_sharded(metric_code, devices)
        nifs.flat_set_reduce_order(self.ref, order)


def test_flat_rs_scripts_on_a_sharded_handle(nifs, oracle_mod):
    """The reference's own flat.rs known-answer scripts (tie-break by id, upsert, delete,
    atomic batch validation, dimension reset, limit edge cases) on a 3-shard resource."""
    for case in load("flat_rs.json"):
        if case.get("differential"):
            continue
        ix = ShardedIndex(nifs, oracle_mod.METRIC_CODE[case["metric"]], [0, 0, 0])
        assert nifs.flat_shard_count(ix.ref) == 3
        run_steps(ix, case["steps"], GpuError)


@pytest.mark.parametrize("metric", [2, 0, 5])
def test_sharded_handle_equals_the_oracle_over_all_rows(nifs, oracle_mod, metric):
    n, d, S = 40_000, 96, 4
    x, ids = make_corpus(n, d, 1200 + metric, metric == 2, oracle_mod, tie_block=40)
    ref = ShardedIndex(nifs, metric, [0] * S)
    route = nifs.flat_route_ids(ref.ref, nifs.pack_ids(ids))
    assert set(route.tolist()) == set(range(S))
    # identical rows that live in different shards: only the id bytes can order them
    first = [int(np.flatnonzero(route == s)[0]) for s in range(S)]
    for r in first[1:]:
        x[r] = x[first[0]]
    unwrap(nifs.flat_load_matrix(ref.ref, ids, x))
    lens = nifs.flat_shard_lens(ref.ref)
    assert sum(lens) == n == len(ref) and min(lens) > n // S * 0.9, lens   # the hash spreads the rows evenly
    assert ref.dimension == d
    packed = oracle_mod.pack_ids(ids)
    rng = np.random.default_rng(3)
    qs = [x[first[0]], x[n // 2]] + [rng.uniform(-1, 1, d).astype(np.float32) for _ in range(3)]
    if metric == 2:
        qs = [oracle_mod.normalize_l2(q) for q in qs]
    for q in qs:
        for limit in (1, 4, 10, 64, 300):
            assert bits(ref.search(q, limit)) == bits(oracle_mod.matrix_search(metric, x, packed, q, limit)), (metric, limit)
    # the first query's top hits are the planted copies, in id-byte order across the shards
    top = ref.search(qs[0], S)
    assert sorted(h[0] for h in top) == sorted(ids[r] for r in first) and [h[0] for h in top] == sorted(h[0] for h in top)
    # limit beyond the row count, and the batch entry point
    assert len(ref.search(qs[1], n + 5)) == n
    batch = np.stack(qs)
    outs = unwrap(nifs.flat_search_batch(ref.ref, batch, 10))
    for q, got in zip(qs, outs):
        assert bits(got) == bits(oracle_mod.matrix_search(metric, x, packed, q, 10))


def test_sharded_mutations_follow_the_oracle(nifs, oracle_mod):
    """Random insert / upsert / delete / search interleaving (flat.rs:59-93) on a 3-shard
    resource: routing, per-shard swap-deletes and rank upkeep must stay invisible."""
    rng = np.random.default_rng(15)
    d = 16
    for m in (0, 2):
        g = ShardedIndex(nifs, m, [0, 0, 0])
        o = oracle_mod.FlatIndex(m)
        live = []
        for step in range(250):
            op = rng.integers(0, 10)
            if op < 5 or not live:
                cnt = int(rng.integers(1, 40))
                items = [("id-%d" % rng.integers(0, 500), rng.uniform(-1, 1, d).astype(np.float32)) for _ in range(cnt)]
                g.insert_many(items)
                o.insert_many(items)
                live = list({*live, *[i for i, _ in items]})
            elif op < 8:
                victim = live.pop(int(rng.integers(0, len(live))))
                g.delete(victim)
                o.delete(victim)
            else:
                g.delete("missing-%d" % step)
                o.delete("missing-%d" % step)
            assert len(g) == len(o) and g.dimension == o.dimension
            if len(o) == 0:
                continue
            q = rng.uniform(-1, 1, d).astype(np.float32)
            k = int(rng.integers(1, 30))
            assert bits(g.search(q, k)) == bits(o.search(q, k)), (m, step)
        # empty it: the resource forgets its dimension (flat.rs:90-92) and takes another one
        for i in list(live):
            g.delete(i)
        assert len(g) == 0 and g.dimension is None
        g.insert("a", [1.0, 2.0, 3.0])
        assert g.dimension == 3
        with pytest.raises(GpuError, match="dimension mismatch"):
            g.insert("b", [1.0, 2.0])
        with pytest.raises(GpuError, match="dimension mismatch"):       # atomic: nothing of the batch lands
            g.insert_many([("c", [1.0, 2.0, 3.0]), ("d", [1.0])])
        assert len(g) == 1


def test_sharded_validation_and_unsupported_calls(nifs, oracle_mod):
    g = ShardedIndex(nifs, 2, [0, 0])
    assert g.search([1.0, 0.0], 0) == [] and g.search([1.0, 0.0], 3) == []
    with pytest.raises(GpuError, match="vector must not be empty"):
        g.search([], 3)
    with pytest.raises(GpuError, match="non-finite"):
        g.search([float("inf"), 0.0], 3)
    g.insert_many([("a", [1.0, 0.0]), ("b", [0.0, 1.0]), ("c", [1.0, 0.0])])
    assert [h[0] for h in g.search([1.0, 0.0], 2)] == [b"a", b"c"]     # vector_algorithms_hardening_test.exs:20-36
    with pytest.raises(GpuError, match="dimension mismatch"):
        g.search([1.0, 0.0, 0.0], 2)
    # collection.ex:276-295 through a two-shard resource
    assert [h[0] for h in unwrap(nifs.flat_quantized_search(g.ref, [1.0, 0.0], 3, 2))] == [b"a", b"c"]
    assert nifs.flat_funnel_search(g.ref, [1.0, 0.0], [3], 3, 2) == ("error", "invalid prefix dimensions")
    assert nifs.flat_funnel_search(g.ref, [1.0, 0.0], [], 3, 2) == ("error", "invalid prefix dimensions")
    assert unwrap(nifs.flat_quantized_search(g.ref, [1.0, 0.0], 0, 2)) == []


@pytest.mark.parametrize("metric", [2, 0, 3])
@pytest.mark.parametrize("rounds", ["one", "per-stage"])
def test_staged_searches_on_a_sharded_handle_equal_the_one_gpu_index(nifs, oracle_mod, metric, rounds, monkeypatch, vt_debug):
    """quantized_search, funnel_search and hybrid_search (collection.ex:276-295, :245-260,
    :325-345) on a 3-shard resource: every step keeps the best rows of a row set, per shard and
    then merged by (rank key, id bytes) -- the results must equal the one-GPU index's (which the
    parity tests pin to the oracle's composition), and the quantized one the oracle's directly.
    Both ways the handle can run them: one fan-out (every shard its whole chain, the handle
    cuts afterwards) and one fan-out per stage (`VT_STAGED_ROUNDS`, also the fallback)."""
    if rounds == "per-stage":
        vt_debug.set("staged_rounds", 1)
    n, d = 60_000, 96     # 20 000 rows per shard: the histogram form of the Hamming pass (n >= 16 384) runs in each
    x, ids = make_corpus(n, d, 2100 + metric, metric == 2, oracle_mod, tie_block=50)
    one = GpuIndex(nifs, metric)
    many = ShardedIndex(nifs, metric, [0, 0, 0])
    unwrap(nifs.flat_load_matrix(one.ref, ids, x))
    unwrap(nifs.flat_load_matrix(many.ref, ids, x))
    rng = np.random.default_rng(6)
    qs = [x[n // 2], x[7]] + [rng.uniform(-1, 1, d).astype(np.float32) for _ in range(3)]
    if metric == 2:
        qs = [oracle_mod.normalize_l2(q) for q in qs]
    by_id = {ids[i]: x[i] for i in range(n)}
    sign = [oracle_mod.compress_sign_bits(r) for r in x]
    for qi, q in enumerate(qs):
        for cand, limit in ((100, 10), (300, 40), (5, 10)):
            got = unwrap(nifs.flat_quantized_search(many.ref, q, cand, limit))
            assert bits(got) == bits(unwrap(nifs.flat_quantized_search(one.ref, q, cand, limit))), (metric, qi, cand)
            if qi < 2 and cand == 100:      # the oracle's own composition: binary_top_k then vector_top_k
                cands = oracle_mod.binary_top_k([(ids[i], sign[i]) for i in range(n)], oracle_mod.compress_sign_bits(q), d, cand)
                want = oracle_mod.vector_top_k([(c, by_id[c]) for c, _ in cands], q, metric, d, limit)
                assert bits(got) == bits(want)
        for stages, cand, limit in (([32], 100, 10), ([16, 48], 64, 10), ([24], 300, 20)):
            got = unwrap(nifs.flat_funnel_search(many.ref, q, stages, cand, limit))
            assert bits(got) == bits(unwrap(nifs.flat_funnel_search(one.ref, q, stages, cand, limit))), (metric, qi, stages)
        gens = [(nifs.GEN_FUNNEL, 50, [32]), (nifs.GEN_QUANTIZED, 60, []), (nifs.GEN_SEARCH, 40, [])]
        got = unwrap(nifs.flat_hybrid_search(many.ref, q, gens, 15))
        assert bits(got) == bits(unwrap(nifs.flat_hybrid_search(one.ref, q, gens, 15))), (metric, qi)
    # the batch forms on both kinds of handle: the one-GPU index groups its queries, the sharded one loops
    qm = np.stack(qs)
    for ref in (one.ref, many.ref):
        got = unwrap(nifs.flat_funnel_search_batch(ref, qm, [16, 48], 64, 10))
        assert [bits(h) for h in got] == [bits(unwrap(nifs.flat_funnel_search(one.ref, q, [16, 48], 64, 10))) for q in qs], metric
        got = unwrap(nifs.flat_quantized_search_batch(ref, qm, 100, 10))
        assert [bits(h) for h in got] == [bits(unwrap(nifs.flat_quantized_search(one.ref, q, 100, 10))) for q in qs], metric
    # after mutations on both (derived columns are patched per shard)
    for target in (one, many):
        target.insert("zz-new", x[7])
        target.delete(ids[8])
    q = qs[1]
    assert bits(unwrap(nifs.flat_quantized_search(many.ref, q, 100, 10))) == bits(unwrap(nifs.flat_quantized_search(one.ref, q, 100, 10)))
    assert bits(unwrap(nifs.flat_funnel_search(many.ref, q, [32], 100, 10))) == bits(unwrap(nifs.flat_funnel_search(one.ref, q, [32], 100, 10)))


@pytest.mark.parametrize("metric", [7, 8])
@pytest.mark.parametrize("rounds", ["one", "per-stage"])
def test_pattern_funnel_on_a_sharded_handle_equals_the_one_gpu_index(nifs, oracle_mod, metric, rounds, monkeypatch, vt_debug):
    """funnel_search / hybrid_search on float hamming / jaccard collections (collection.ex:245-260, :325-345): the
    first stage of each shard reads the prefix of its non-zero-bit column.  Equal to the one-GPU index (pinned to the
    oracle's composition in test_gpu_parity) single, batched, and after mutations patched the column."""
    if rounds == "per-stage":
        vt_debug.set("staged_rounds", 1)
    n, d = 60_000, 130
    rng = np.random.default_rng(77 + metric)
    x = (rng.uniform(-1, 1, (n, d)) * (rng.uniform(0, 1, (n, d)) < 0.4)).astype(np.float32)
    x[100:140] = x[100]
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    one = GpuIndex(nifs, metric)
    many = ShardedIndex(nifs, metric, [0, 0, 0])
    unwrap(nifs.flat_load_matrix(one.ref, ids, x))
    unwrap(nifs.flat_load_matrix(many.ref, ids, x))
    qs = [x[100], x[n - 1]] + [(rng.uniform(-1, 1, d) * (rng.uniform(0, 1, d) < 0.4)).astype(np.float32) for _ in range(3)]
    for qi, q in enumerate(qs):
        for stages, cand, limit in (([64], 100, 10), ([70, 129], 64, 10), ([24], 300, 20), ([130], 20, 20)):
            got = unwrap(nifs.flat_funnel_search(many.ref, q, stages, cand, limit))
            assert bits(got) == bits(unwrap(nifs.flat_funnel_search(one.ref, q, stages, cand, limit))), (metric, qi, stages)
        gens = [(nifs.GEN_FUNNEL, 50, [65]), (nifs.GEN_SEARCH, 40, [])]
        got = unwrap(nifs.flat_hybrid_search(many.ref, q, gens, 15))
        assert bits(got) == bits(unwrap(nifs.flat_hybrid_search(one.ref, q, gens, 15))), (metric, qi)
    qm = np.stack(qs)
    for ref in (one.ref, many.ref):
        got = unwrap(nifs.flat_funnel_search_batch(ref, qm, [70, 129], 64, 10))
        assert [bits(h) for h in got] == [bits(unwrap(nifs.flat_funnel_search(one.ref, q, [70, 129], 64, 10))) for q in qs], metric
    by_id = {ids[i]: x[i] for i in range(n)}
    newrow = x[n - 1].copy()
    for target in (one, many):
        target.insert("zz-new", newrow)
        target.delete(ids[101])
    by_id[b"zz-new"] = newrow
    del by_id[ids[101]]
    rows = list(by_id.items())
    q = qs[1]
    kept = oracle_mod.vector_top_k(rows, q, metric, 70, 100)
    want = oracle_mod.vector_top_k([(i, by_id[i]) for i, _ in kept], q, metric, d, 10)
    assert bits(unwrap(nifs.flat_funnel_search(one.ref, q, [70], 100, 10))) == bits(want)
    assert bits(unwrap(nifs.flat_funnel_search(many.ref, q, [70], 100, 10))) == bits(want)


def test_one_round_staged_search_ignores_an_overflow_outside_the_candidate_set(nifs, oracle_mod):
    """The one-round form reranks every shard's OWN candidates; one of them may be a row the
    handle-wide candidate set does not contain.  If that row's full-length distance overflows
    (here: L2 over components of 3e38, beyond f32 even through the f64 recovery,
    distances.rs:70-98), the reference -- which never looks at it -- answers normally, and so
    must the handle (it redoes the call stage by stage).  A row inside the set still raises."""
    n, d, S = 3000, 16, 3
    rng = np.random.default_rng(12)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    ids = [b"doc-%04d" % i for i in range(n)]
    many = ShardedIndex(nifs, 0, [0] * S)
    one = GpuIndex(nifs, 0)
    route = nifs.flat_route_ids(many.ref, nifs.pack_ids(ids))
    q = rng.uniform(-1, 1, d).astype(np.float32)
    on0 = np.flatnonzero(route == 0)
    great = list(np.flatnonzero(route == 1)[:3]) + list(np.flatnonzero(route == 2)[:3])
    for r in great:                       # six rows whose 8-float prefix IS the query's: the whole candidate set (5)
        x[r, :8] = q[:8]
    bad = int(on0[0])                     # shard 0's best row under the prefix, 7th overall
    x[bad, :8] = q[:8] + np.float32(0.001)
    x[bad, 8:] = np.float32(3e38)
    for ix in (one, many):
        unwrap(nifs.flat_load_matrix(ix.ref, ids, x))
    want = unwrap(nifs.flat_funnel_search(one.ref, q, [8], 5, 3))
    assert bits(unwrap(nifs.flat_funnel_search(many.ref, q, [8], 5, 3))) == bits(want)
    assert {h[0] for h in want} <= {ids[r] for r in great}
    gens = [(nifs.GEN_FUNNEL, 5, [8])]
    assert bits(unwrap(nifs.flat_hybrid_search(many.ref, q, gens, 3))) == bits(unwrap(nifs.flat_hybrid_search(one.ref, q, gens, 3)))
    # seven candidates: the row is in the set, the rerank meets the overflow in both
    assert nifs.flat_funnel_search(one.ref, q, [8], 7, 3) == ("error", "metric overflow")
    assert nifs.flat_funnel_search(many.ref, q, [8], 7, 3) == ("error", "metric overflow")


def test_rccl_exchange_with_a_one_rank_communicator(nifs, oracle_mod, monkeypatch, vt_debug):
    """The RCCL leg of the exchange (librccl loaded at run time, ncclCommInitAll, one ncclAllGather
    per shard queued behind the scan, gathered lists copied out by shard 0) on the only GPU of
    the box: a one-shard resource forced onto the worker path."""
    vt_debug.set("shard_force_workers", 1)
    vt_debug.set("shard_exchange", 2)
    n, d = 30_000, 128
    x, ids = make_corpus(n, d, 77, True, oracle_mod, tie_block=30)
    g = ShardedIndex(nifs, 2, [0])
    from vettore_amd import _lib
    assert nifs.flat_exchange(g.ref) == _lib.EXCHANGE_RCCL and nifs.flat_rccl_ranks(g.ref) == 1
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    packed = oracle_mod.pack_ids(ids)
    rng = np.random.default_rng(8)
    for i in range(6):
        q = x[n // 2] if i == 0 else oracle_mod.normalize_l2(rng.uniform(-1, 1, d).astype(np.float32))
        for limit in (10, 200, 300):      # 300 > 256: served by the host exchange
            assert bits(g.search(q, limit)) == bits(oracle_mod.matrix_search(2, x, packed, q, limit)), (i, limit)
    # unsorted inserts, then the exchange path again (ranks are rebuilt strictly for it)
    g.insert("aaa-first", x[3])
    g.delete(ids[10])
    keep = [i for i in range(n) if i != 10]
    x2 = np.concatenate([x[keep], x[3:4]])
    ids2 = [ids[i] for i in keep] + [b"aaa-first"]
    assert bits(g.search(x[3], 5)) == bits(oracle_mod.matrix_search(2, x2, oracle_mod.pack_ids(ids2), x[3], 5))
    # two shards on ONE device cannot form a communicator: the handle says so and stays on the host path
    vt_debug.reset("shard_exchange")
    two = ShardedIndex(nifs, 2, [0, 0])
    assert nifs.flat_exchange(two.ref) == _lib.EXCHANGE_HOST and nifs.flat_rccl_ranks(two.ref) == 0
    res = nifs.flat_set_exchange(two.ref, _lib.EXCHANGE_RCCL)
    assert res[0] == "error" and "own device" in res[1]
    # ... and each handle remembers what it chose and why (vt_flat_exchange_note)
    assert "RCCL all-gather" in nifs.flat_exchange_note(g.ref) and "1 ranks" in nifs.flat_exchange_note(g.ref)
    assert "share a device" in nifs.flat_exchange_note(two.ref)


def test_a_wedged_exchange_times_out_with_a_message(nifs, oracle_mod, monkeypatch, request, vt_debug):
    """The first 8-GPU run must not hang without a word if a collective never completes (VERDICT r2
    next #7a): the wait behind a shard's all-gather has a deadline (VT_EXCHANGE_TIMEOUT_MS); past it
    the search fails with "RCCL exchange timed out on shard s ...", the handle is poisoned (its
    stream still holds the stuck collective).  The stall is injected in front of the all-gather
    (VT_TEST_EXCHANGE_STALL_MS, libvettore_hip_hooks.so only: the test re-runs itself there)."""
    if support.rerun_with_hooks_library(request):
        return
    import time
    vt_debug.set("shard_force_workers", 1)
    vt_debug.set("shard_exchange", 2)
    vt_debug.set("exchange_timeout_ms", 250)     # (read once per process: before the first exchange)
    g = ShardedIndex(nifs, 0, [0])
    g.insert_many([("a", [0.0, 0.0]), ("b", [1.0, 0.0]), ("c", [2.0, 0.0])])
    assert [h[0] for h in g.search([0.9, 0.0], 2)] == [b"b", b"a"]
    vt_debug.set("test_exchange_stall_ms", 1500)
    t0 = time.perf_counter()
    res = nifs.flat_search(g.ref, [0.9, 0.0], 2)
    waited = time.perf_counter() - t0
    assert res[0] == "error" and "RCCL exchange timed out on shard 0" in res[1], res
    assert 0.2 < waited < 1.45, waited                      # (after the deadline, before the 1.5-s stall ends: the deadline cut it)
    vt_debug.reset("test_exchange_stall_ms")
    assert nifs.flat_search(g.ref, [0.9, 0.0], 2) == ("error", "flat lock poisoned")
    assert nifs.flat_insert(g.ref, "d", [3.0, 0.0]) == ("error", "flat lock poisoned")
    time.sleep(1.6)                                         # the injected stall ends; the stream drains before the handle goes
    fresh = ShardedIndex(nifs, 0, [0])                      # other handles (and their communicators) are unaffected
    fresh.insert("x", [1.0, 1.0])
    assert fresh.search([1.0, 1.0], 1) == [(b"x", 0.0)]


def test_concurrent_searches_over_the_rccl_exchange_keep_their_own_lists(nifs, oracle_mod, monkeypatch, vt_debug):
    """Searches run under the shared lock, so two callers can be in the RCCL leg at once: their jobs
    pass through every shard's worker in one order, but each caller merges on its own thread --
    from its own copy of the gathered lists (shard 0's pinned buffer is refilled by the next job).
    Eight threads, distinct queries, every answer checked."""
    vt_debug.set("shard_force_workers", 1)
    vt_debug.set("shard_exchange", 2)
    vt_debug.set("coalesce", 0)          # every caller runs its own search_multi
    n, d = 20_000, 64
    x, ids = make_corpus(n, d, 78, True, oracle_mod, tie_block=30)
    g = ShardedIndex(nifs, 2, [0])
    assert nifs.flat_rccl_ranks(g.ref) == 1
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(3)
    qs = [x[n // 2]] + [oracle_mod.normalize_l2(rng.uniform(-1, 1, d).astype(np.float32)) for _ in range(23)]
    want = [bits(g.search(q, 10)) for q in qs]
    bad = []

    def worker(t):
        for r in range(150):
            j = (t * 5 + r) % len(qs)
            if bits(g.search(qs[j], 10)) != want[j]:
                bad.append((t, r, j))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not bad, bad[:3]


def test_stale_external_ranks_are_reported_to_every_rank(nifs, oracle_mod):
    """after vt_flat_set_id_ranks a delete moves rows under the caller's id table and
    an insert drops the installed ranks.  vt_flat_search_begin then marks its block instead of
    scanning, and vt_flat_merge_gathered -- which every rank runs on the same gathered blocks --
    fails with 'stale id ranks', so all ranks fall back together."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n, d = 5000, 32
    x, ids = make_corpus(n, d, 5, False, oracle_mod)
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    assert nifs.flat_set_id_ranks(g.ref, nifs.rank_ids(nifs.pack_ids(ids))) == "ok"
    block = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(block), 16 + 16 * 10) == 0
    bufs = nifs.MergeBuffers()
    assert nifs.flat_search_begin(g.ref, x[0], 10, block.value) == "ok"
    st, cnt = nifs.flat_merge_gathered(g.ref, block.value, 1, 10, 176, bufs)
    assert st == "ok" and cnt == 10
    g.insert(ids[7], x[8])                     # an upsert moves nothing: still valid
    assert nifs.flat_search_begin(g.ref, x[0], 10, block.value) == "ok"
    assert nifs.flat_merge_gathered(g.ref, block.value, 1, 10, 176, bufs)[0] == "ok"
    g.delete(ids[3])                           # the last row moves into slot 3
    assert nifs.flat_search_begin(g.ref, x[0], 10, block.value) == "ok"
    res = nifs.flat_merge_gathered(g.ref, block.value, 1, 10, 176, bufs)
    assert res[0] == "error" and "stale id ranks" in res[1]
    # the plain search is unaffected
    keep = [i for i in range(n) if i != 3]
    x2 = x[keep].copy()
    x2[keep.index(7)] = x[8]
    assert bits(g.search(x[0], 10)) == bits(oracle_mod.matrix_search(0, x2, oracle_mod.pack_ids([ids[i] for i in keep]), x[0], 10))
    assert hip.hipFree(block) == 0


def test_many_readers_and_a_writer_on_one_handle(nifs, oracle_mod):
    """nifs.rs:297-309 takes the read lock, :259-295 the write lock: eight reader threads search
    ONE handle (each on its own stream and scratch) while a writer inserts, upserts and deletes
    rows that can never reach a top-10 (far from every query); every answer equals the oracle's."""
    n, d = 20_000, 64
    x, ids = make_corpus(n, d, 321, False, oracle_mod, tie_block=16)
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    packed = oracle_mod.pack_ids(ids)
    rng = np.random.default_rng(2)
    qs = [x[n // 2]] + [rng.uniform(-1, 1, d).astype(np.float32) for _ in range(7)]
    want = [bits(oracle_mod.matrix_search(0, x, packed, q, 10)) for q in qs]
    far = (rng.uniform(-1, 1, (400, d)) + 50.0).astype(np.float32)
    errors, stop = [], threading.Event()

    def reader(t):
        try:
            for i in range(200):
                j = (t + i) % len(qs)
                mode = (t + i) % 3
                if mode == 0:
                    got = bits(unwrap(nifs.flat_search(g.ref, qs[j], 10)))
                elif mode == 1:
                    got = bits(unwrap(nifs.flat_search_batch(g.ref, np.stack([qs[j], qs[j]]), 10))[1])
                else:
                    got = bits(unwrap(nifs.flat_search(g.ref, qs[j], 300))[:10])
                if got != want[j]:
                    errors.append(("reader", t, i, mode))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(("reader", t, repr(e)))

    def writer():
        try:
            i = 0
            while not stop.is_set():
                key = "zz-far-%d" % (i % 400)
                g.insert(key, far[i % 400])
                if i % 3 == 2:
                    g.delete("zz-far-%d" % ((i - 2) % 400))
                if i % 7 == 0:
                    g.insert_many([("aa-far-%d" % (i % 50), far[(i * 3) % 400])])   # sorts in front of every "doc-"
                i += 1
        except Exception as e:  # noqa: BLE001
            errors.append(("writer", repr(e)))

    ths = [threading.Thread(target=reader, args=(t,)) for t in range(8)]
    w = threading.Thread(target=writer)
    w.start()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    stop.set()
    w.join()
    assert not errors, errors[:3]
    # quantized / funnel readers next to each other on a second handle (cosine)
    xc, idc = make_corpus(20_000, 64, 99, True, oracle_mod)
    c = GpuIndex(nifs, 2)
    unwrap(nifs.flat_load_matrix(c.ref, idc, xc))
    qn = [oracle_mod.normalize_l2(q) for q in qs]
    wq = [bits(unwrap(nifs.flat_quantized_search(c.ref, q, 100, 10))) for q in qn]
    wf = [bits(unwrap(nifs.flat_funnel_search(c.ref, q, [32], 100, 10))) for q in qn]

    def reader2(t):
        try:
            for i in range(60):
                j = (t + i) % len(qn)
                if (t + i) % 2:
                    ok = bits(unwrap(nifs.flat_quantized_search(c.ref, qn[j], 100, 10))) == wq[j]
                else:
                    ok = bits(unwrap(nifs.flat_funnel_search(c.ref, qn[j], [32], 100, 10))) == wf[j]
                if not ok:
                    errors.append(("reader2", t, i))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(("reader2", t, repr(e)))

    ths = [threading.Thread(target=reader2, args=(t,)) for t in range(6)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors[:3]


def test_bench_runs_as_the_driver_invokes_it_for_two_gpus():
    """`python bench.py --gpus 2` with no launcher: one process, one handle over
    two shards -- both on device 0 here (--devices 0,0), the box has one GPU -- and ONE JSON line
    last, n_gpus = 2."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--devices", "0,0", "--rows", "200000",
                          "--dim", "64", "--steps", "20", "--warmup", "3", "--no-cpu"], capture_output=True, text=True,
                         env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["value"] > 0
    assert "one handle over 2 devices" in line["config"]["sharding"] and line["config"]["processes"] == 1
    assert line["roofline"]["algorithmic_bytes_per_launch"] == line["config"]["rows_per_gpu"] * 64 * 4 or \
        abs(line["roofline"]["algorithmic_bytes_per_launch"] - 100000 * 64 * 4) < 0.05 * 100000 * 64 * 4


@pytest.mark.parametrize("devices", [[0], [0, 0]])
def test_a_mutation_that_dies_half_way_poisons_the_handle(nifs, oracle_mod, devices, monkeypatch, request, vt_debug):
    """nifs.rs:266-309: a panic under the write lock poisons the RwLock and every later NIF call
    returns {:error, "flat lock poisoned"}.  Here: a device failure after a mutation began changing
    the index (injected between the id table's update and the rows' arrival).  Validation errors
    come before anything is stored (flat.rs:69-85) and poison nothing.  (The injection hook only
    exists in libvettore_hip_hooks.so: the test re-runs itself in a process that loads that build.)"""
    if support.rerun_with_hooks_library(request):
        return
    g = ShardedIndex(nifs, 0, devices)
    g.insert_many([("a", [0.0, 0.0]), ("b", [1.0, 0.0]), ("c", [2.0, 0.0])])
    with pytest.raises(GpuError, match="dimension mismatch"):
        g.insert("d", [1.0])
    with pytest.raises(GpuError, match="non-finite"):
        g.insert_many([("d", [1.0, 1.0]), ("e", [float("nan"), 0.0])])
    assert [h[0] for h in g.search([0.9, 0.0], 2)] == [b"b", b"a"] and len(g) == 3      # still healthy
    vt_debug.set("test_fail_after_id_update", 1)
    res = nifs.flat_insert(g.ref, "d", [3.0, 0.0])
    assert res[0] == "error" and "injected" in res[1]
    vt_debug.reset("test_fail_after_id_update")
    for res in (nifs.flat_search(g.ref, [0.9, 0.0], 2), nifs.flat_insert(g.ref, "e", [4.0, 0.0]),
                nifs.flat_delete(g.ref, "a"), nifs.flat_search_batch(g.ref, np.zeros((2, 2), np.float32), 1),
                nifs.flat_quantized_search(g.ref, [0.9, 0.0], 3, 2)):
        assert res == ("error", "flat lock poisoned"), res
    fresh = ShardedIndex(nifs, 0, devices)       # other handles are unaffected
    fresh.insert("x", [1.0, 1.0])
    assert fresh.search([1.0, 1.0], 1) == [(b"x", 0.0)]


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_device_resident_batches_with_scattered_rows(nifs, oracle_mod, devices):
    """vt_flat_load_device_matrix when the batch does not land as one block: a batch dealt to
    shards by the hash of its ids, upserts of rows all over the slab, an id twice in one batch
    (the last occurrence wins, flat.rs:270-281) -- one gather launch per shard, same index as
    the oracle's."""
    import torch
    n, d = 20_000, 40
    x, ids = make_corpus(n, d, 31, False, oracle_mod)
    g = ShardedIndex(nifs, 0, devices)
    o = oracle_mod.FlatIndex(0)
    xd = torch.from_numpy(x).to("cuda:0")
    assert nifs.flat_load_device_matrix(g.ref, nifs.pack_ids(ids), xd.data_ptr(), n, d) == ("ok", ())
    o.insert_matrix(ids, x)
    # upserts of every third row in shuffled order, one id twice, plus new ids, from device memory
    rng = np.random.default_rng(5)
    pick = rng.permutation(np.arange(0, n, 3))[:3000]
    up_ids = [ids[i] for i in pick] + [ids[int(pick[0])]] + [b"new-%d" % i for i in range(200)]
    up = rng.uniform(-1, 1, (len(up_ids), d)).astype(np.float32)
    ud = torch.from_numpy(up).to("cuda:0")
    assert nifs.flat_load_device_matrix(g.ref, nifs.pack_ids(up_ids), ud.data_ptr(), len(up_ids), d) == ("ok", ())
    o.insert_many(list(zip(up_ids, up)))
    assert len(g) == len(o) == n + 200
    for q in (up[0], up[len(pick)], up[-1], x[1], rng.uniform(-1, 1, d).astype(np.float32)):
        assert bits(g.search(q, 20)) == bits(o.search(q, 20))
    # the row written twice holds its LAST value
    assert g.search(up[len(pick)], 1)[0] == (ids[int(pick[0])], 0.0)


@pytest.mark.parametrize("metric", [7, 8])
def test_pattern_metrics_on_a_sharded_handle(nifs, oracle_mod, metric):
    """Float hamming / jaccard on two shards of 16 384 rows or more each: every shard answers from its
    own column of non-zero bits (K4 / K4p on the shard's worker), the lists meet by (rank key, id
    bytes); single searches, batches, after mutations -- the oracle's hits over all rows, bit for bit."""
    n, d, S = 44_000, 128, 2
    rng = np.random.default_rng(1700 + metric)
    x = (rng.uniform(-1, 1, (n + 60, d)) * (rng.uniform(0, 1, (n + 60, d)) < 0.35)).astype(np.float32)
    ids = [b"doc-%05d" % i for i in range(n + 60)]
    ref = ShardedIndex(nifs, metric, [0] * S)
    unwrap(nifs.flat_load_matrix(ref.ref, ids[:n], x[:n]))
    assert min(nifs.flat_shard_lens(ref.ref)) >= 16_384
    cur = {ids[i]: x[i] for i in range(n)}
    nifs.flat_set_profiling(ref.ref, True)

    def check():
        keys = sorted(cur)
        mat = np.stack([cur[k] for k in keys])
        packed = oracle_mod.pack_ids(keys)
        qs = np.stack([x[3], np.zeros(d, np.float32), x[n // 2], x[77], x[78]])
        nifs.flat_get_profile(ref.ref, reset=True)
        for q in qs[:3]:
            for limit in (1, 10, 300):
                got = unwrap(nifs.flat_search(ref.ref, q, limit))
                assert bits(got) == bits(oracle_mod.matrix_search(metric, mat, packed, q, limit))
        got = unwrap(nifs.flat_search_batch(ref.ref, qs, 10))
        for i in range(len(qs)):
            assert bits(got[i]) == bits(oracle_mod.matrix_search(metric, mat, packed, qs[i], 10))
        prof = nifs.flat_get_profile(ref.ref, reset=True)
        assert prof["hamming_launches"] >= 1 and prof["scan_launches"] == 0, prof

    check()
    for i in range(n + 59, n - 1, -1):                    # appends in descending id order
        ref.insert(ids[i], x[i]); cur[ids[i]] = x[i]
    for victim in (ids[5], ids[n - 3]):
        ref.delete(victim); cur.pop(victim, None)
    v = np.where(cur[ids[9]] == 0, np.float32(1.5), np.float32(0.0)).astype(np.float32)
    ref.insert(ids[9], v); cur[ids[9]] = v
    check()


@pytest.mark.parametrize("metric", [7, 8])
def test_pattern_metric_readers_and_a_writer_on_one_handle(nifs, oracle_mod, metric):
    """The same under float hamming / jaccard: readers answer from the non-zero-bit column (singly on
    their own contexts, or as K4p batches when they meet) while a writer inserts, upserts and deletes
    rows -- every one of which dirties the column -- that can never reach a top-10; every answer equals
    the oracle's."""
    n, d = 20_000, 64
    rng = np.random.default_rng(1300 + metric)
    x = (rng.uniform(-1, 1, (n, d)) * (rng.uniform(0, 1, (n, d)) < 0.3)).astype(np.float32)
    ids = [b"doc-%05d" % i for i in range(n)]
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    packed = oracle_mod.pack_ids(ids)
    qs = [x[n // 2]] + [(rng.uniform(-1, 1, d) * (rng.uniform(0, 1, d) < 0.3)).astype(np.float32) for _ in range(7)]
    want = [bits(oracle_mod.matrix_search(metric, x, packed, q, 10)) for q in qs]
    # far from every query: all coordinates non-zero under hamming (distance >= 64 - |q|); one non-zero
    # coordinate under jaccard (distance >= 1 - 1 / |q|, where random rows reach ~0.5)
    far = rng.uniform(0.5, 1.0, (400, d)).astype(np.float32)
    if metric == 8:
        far *= np.eye(d, dtype=np.float32)[rng.integers(0, d, 400)]
    errors, stop = [], threading.Event()

    def reader(t):
        try:
            for i in range(150):
                j = (t + i) % len(qs)
                mode = (t + i) % 3
                if mode == 0:
                    got = bits(unwrap(nifs.flat_search(g.ref, qs[j], 10)))
                elif mode == 1:
                    got = bits(unwrap(nifs.flat_search_batch(g.ref, np.stack([qs[j], qs[j]]), 10))[1])
                else:
                    got = bits(unwrap(nifs.flat_search(g.ref, qs[j], 100))[:10])
                if got != want[j]:
                    errors.append(("reader", t, i, mode))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(("reader", t, repr(e)))

    def writer():
        try:
            i = 0
            while not stop.is_set():
                g.insert("zz-far-%d" % (i % 400), far[i % 400])
                if i % 3 == 2:
                    g.delete("zz-far-%d" % ((i - 2) % 400))
                if i % 7 == 0:
                    g.insert_many([("aa-far-%d" % (i % 50), far[(i * 3) % 400])])   # sorts in front of every "doc-"
                i += 1
        except Exception as e:  # noqa: BLE001
            errors.append(("writer", repr(e)))

    ths = [threading.Thread(target=reader, args=(t,)) for t in range(8)]
    w = threading.Thread(target=writer)
    w.start()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    stop.set()
    w.join()
    assert not errors, errors[:3]


@pytest.mark.parametrize("metric,shards,nq", [(0, 3, 700), (1, 2, 1100), (2, 4, 520)])
def test_long_batches_on_a_sharded_handle_merge_while_the_shards_still_run(nifs, oracle_mod, vt_debug, metric, shards, nq):
    """A batch of several 256-query groups on a sharded handle (round 5): every shard settles its lists group by group,
    the calling thread merges a query as soon as every shard has settled it -- under the later groups' passes -- and
    what is left when the shards are through is merged afterwards.  Whatever the timing, every list is the one-GPU
    index's (itself checked against the oracle here) and the query's own single search on the sharded handle, bit for
    bit; the planted duplicates sit in different shards, so their order is the id bytes' across shards."""
    vt_debug.set("force_batch_mfma", 1)    # (the cost model would answer corpora this small with sweeps)
    n, d = 30_000, 128
    x, ids = make_corpus(n, d, 7700 + metric, metric == 2, oracle_mod, tie_block=30)
    sharded = ShardedIndex(nifs, metric, [0] * shards)
    route = nifs.flat_route_ids(sharded.ref, nifs.pack_ids(ids))
    first = [int(np.flatnonzero(route == s)[0]) for s in range(shards)]
    for r in first[1:]:
        x[r] = x[first[0]]
    unwrap(nifs.flat_load_matrix(sharded.ref, ids, x))
    plain = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(plain.ref, ids, x))
    rng = np.random.default_rng(91 + metric)
    qs = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    qs[0], qs[255], qs[256], qs[nq - 1] = x[first[0]], x[first[-1]], x[n // 2], x[first[0]]
    if metric == 2:
        qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
    packed = oracle_mod.pack_ids(ids)
    for limit in (10, 3):
        got = unwrap(nifs.flat_search_batch(sharded.ref, qs, limit))
        one = unwrap(nifs.flat_search_batch(plain.ref, qs, limit))
        assert len(got) == nq
        for i in range(nq):
            assert bits(got[i]) == bits(one[i]), (metric, limit, i)
        for i in (0, 255, 256, 257, 511, 512, nq - 1):
            if i < nq:
                assert bits(got[i]) == bits(sharded.search(qs[i], limit)), (metric, limit, i)
                assert bits(got[i]) == bits(oracle_mod.matrix_search(metric, x, packed, qs[i], limit)), (metric, limit, i)
    # the duplicates planted across the shards come back in id order
    top = unwrap(nifs.flat_search_batch(sharded.ref, qs, shards))[0]
    assert [h[0] for h in top] == sorted(ids[r] for r in first)


def test_eight_shards_answer_4096_queries_in_one_call(nifs, oracle_mod, vt_debug):
    """The width of the node on one card: vt_flat_new_sharded(metric, [0] * 8), 200 000 rows, 4 096 queries in ONE call --
    sixteen groups of 256 per shard, the calling thread merging every query as soon as all eight shards have settled it
    (run_on_workers_meanwhile, host/vt_multi.h).  A row whose f32 products overflow against the queries of a MIDDLE group
    (distances.rs:61-68: recomputed in f64, representable again) sits in one shard; every list equals the one-GPU index's,
    a sample the oracle's, bit for bit.  Then one query of that group is made to overflow for good: the call reports
    "metric overflow" as that query's own search does, and the handle answers the next call as before."""
    vt_debug.set("force_batch_mfma", 1)
    metric, n, d, shards, nq = 3, 200_000, 64, 8, 4096
    x, ids = make_corpus(n, d, 8800, False, oracle_mod, tie_block=40)
    x = x.copy()
    sharded = ShardedIndex(nifs, metric, [0] * shards)
    assert nifs.flat_shard_count(sharded.ref) == shards
    route = nifs.flat_route_ids(sharded.ref, nifs.pack_ids(ids))
    assert set(route.tolist()) == set(range(shards))
    big = np.finfo(np.float32).max
    hot = int(np.flatnonzero(route == 5)[1000])           # a row in the middle of shard 5
    x[hot, 0] = x[hot, 1] = big
    x[hot, 2:] = 4.0                                       # (the rest of its dot product makes it every middle query's best hit or worst)
    rng = np.random.default_rng(88)
    qs = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    qs[:, 0:2] = 0.0
    middle = range(1900, 2200)                             # queries 1900..2199: groups 7 and 8 of 16
    for i in middle:
        qs[i, 0], qs[i, 1] = 2.0, -2.0                     # 2 * max - 2 * max: inf - inf in f32, 0 in f64
    first = [int(np.flatnonzero(route == s)[0]) for s in range(shards)]
    for r in first[1:]:
        x[r] = x[first[0]]                                 # identical rows in all eight shards: the id bytes order them
    qs[0], qs[4095] = x[first[0]], x[first[3]]
    qs[0, 0:2] = qs[4095, 0:2] = 0.0
    unwrap(nifs.flat_load_matrix(sharded.ref, ids, x))
    plain = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(plain.ref, ids, x))
    packed = oracle_mod.pack_ids(ids)
    got = unwrap(nifs.flat_search_batch(sharded.ref, qs, 10))
    one = unwrap(nifs.flat_search_batch(plain.ref, qs, 10))
    assert len(got) == nq
    for i in range(nq):
        assert bits(got[i]) == bits(one[i]), i
    sample = [0, 1, 255, 256, 1899, 1900, 1901, 2047, 2048, 2199, 2200, 4095] + [int(v) for v in rng.integers(0, nq, 60)]
    for i in sample:
        want = oracle_mod.matrix_search(metric, x, packed, qs[i], 10)
        assert bits(got[i]) == bits(want), i
        assert bits(sharded.search(qs[i], 10)) == bits(want), i
    # the overflowing row was recomputed, not dropped: it is the best hit of many a middle query
    assert np.isfinite(oracle_mod.compute(metric, qs[2000], x[hot]))
    assert sum(1 for i in middle if got[i][0][0] == ids[hot]) >= 30
    # the identical rows planted in all eight shards score alike: they come back side by side, in id order across the shards
    planted = {ids[r] for r in first}
    wide = [h[0] for h in unwrap(nifs.flat_search_batch(sharded.ref, qs[:1], 64))[0]]
    at = [i for i, key in enumerate(wide) if key in planted]
    assert len(at) == shards and at == list(range(at[0], at[0] + shards)), at
    assert [wide[i] for i in at] == sorted(planted)
    # one query of the middle group overflows for good
    bad = qs.copy()
    bad[2000, 0] = bad[2000, 1] = 2.0
    with pytest.raises(oracle_mod.OracleError, match="metric overflow"):
        oracle_mod.matrix_search(metric, x, packed, bad[2000], 10)
    assert nifs.flat_search(sharded.ref, bad[2000], 10) == ("error", "metric overflow")
    assert nifs.flat_search_batch(sharded.ref, bad, 10) == ("error", "metric overflow")
    assert nifs.flat_search_batch(plain.ref, bad, 10) == ("error", "metric overflow")
    again = unwrap(nifs.flat_search_batch(sharded.ref, qs[1800:2400], 10))
    for j, i in enumerate(range(1800, 2400)):
        assert bits(again[j]) == bits(one[i]), i
