"""K1m (vt_scan_multi.hip): several queries per sweep of the corpus, exact arithmetic, all nine
metrics -- `-m gpu`.  Every query of a batch must get the hits its own flat_search gets, bit for
bit (ids, order, raw), which the oracle pins in turn."""
import numpy as np
import pytest

from test_gpu_parity import GpuError, GpuIndex, bits, make_corpus, nifs, unwrap  # noqa: F401  (nifs is a fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def no_matrix_cores(monkeypatch, vt_debug):
    # dot-family batches would otherwise take the shared MFMA pass: here K1m serves every metric
    vt_debug.set("batch_no_mfma", 1)
    # ... and on corpora of a few thousand rows two or three single scans would be priced lower than a sweep
    vt_debug.set("force_multi_scan", 1)


@pytest.mark.parametrize("metric", range(9))
def test_multi_query_scan_equals_single_queries_all_metrics(nifs, oracle_mod, metric):
    rng = np.random.default_rng(40 + metric)
    # (64, 128, 320, 384, 640: the last panel of a row is 64 or 128 floats -- the builds that pack it
    # for four or two rows into one load; 128 and 64 are that panel alone)
    for d in (7, 24, 64, 100, 128, 192, 256, 320, 384, 448, 640, 1000):
        n = 6000 if d >= 384 else 12000
        x, ids = make_corpus(n, d, 700 + metric + d, metric == 2, oracle_mod, tie_block=40)
        if metric in (7, 8):                       # float hamming / jaccard: zeros must occur
            x[rng.uniform(size=x.shape) < 0.3] = 0.0
        packed = oracle_mod.pack_ids(ids)
        for order in ((3,) if d != 100 else (0, 1, 2, 3)):
            oracle_mod.set_reduce_order(order)
            try:
                g = GpuIndex(nifs, metric, order)
                unwrap(nifs.flat_load_matrix(g.ref, ids, x))
                nifs.flat_set_profiling(g.ref, True)
                for nq, k in ((2, 10), (3, 1), (5, 32), (8, 10), (9, 7), (17, 10)):
                    qs = rng.uniform(-1, 1, size=(nq, d)).astype(np.float32)
                    qs[0] = x[n // 2]                  # sits on the block of identical rows
                    if metric in (7, 8):
                        qs[rng.uniform(size=qs.shape) < 0.3] = 0.0
                    if metric == 2:
                        qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
                    nifs.flat_get_profile(g.ref, reset=True)
                    got = unwrap(nifs.flat_search_batch(g.ref, qs, k))
                    prof = nifs.flat_get_profile(g.ref, reset=True)
                    assert prof["scan_launches"] == (nq + 7) // 8, (prof, nq)   # sweeps, not nq scans
                    for i in range(nq):
                        assert bits(got[i]) == bits(oracle_mod.matrix_search(metric, x, packed, qs[i], k)), (metric, d, order, nq, k, i)
                # a limit the small wave buffers cannot hold goes query by query, same answers
                qs = rng.uniform(-1, 1, size=(3, d)).astype(np.float32)
                if metric == 2:
                    qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
                got = unwrap(nifs.flat_search_batch(g.ref, qs, 40))
                for i in range(3):
                    assert bits(got[i]) == bits(oracle_mod.matrix_search(metric, x, packed, qs[i], 40))
            finally:
                oracle_mod.set_reduce_order(oracle_mod.DEFAULT_ORDER)


def test_multi_query_scan_overflow_and_recovery(nifs, oracle_mod):
    """distances.rs:59-67 inside a sweep: an f32 sum that overflows is recomputed in f64 for
    that (query, row) pair alone; one that stays unrepresentable fails the call with 'metric
    overflow', as the query's own flat_search would."""
    n, d = 3000, 40
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    x[17] = 1e20                                       # L2 to an ordinary query: sum of squares overflows f32, sqrt does not
    ids = [b"r%04d" % i for i in range(n)]
    packed = oracle_mod.pack_ids(ids)
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    qs = rng.uniform(-1, 1, (6, d)).astype(np.float32)
    got = unwrap(nifs.flat_search_batch(g.ref, qs, n)) if False else unwrap(nifs.flat_search_batch(g.ref, qs, 32))
    for i in range(6):
        assert bits(got[i]) == bits(oracle_mod.matrix_search(0, x, packed, qs[i], 32))
    far = np.full(d, -1e20, np.float32)                # this query makes row 17's distance recoverable and LARGE: it is last
    qs2 = np.stack([qs[0], far, qs[1]])
    got = unwrap(nifs.flat_search_batch(g.ref, qs2, 5))
    for i in range(3):
        assert bits(got[i]) == bits(oracle_mod.matrix_search(0, x, packed, qs2[i], 5))
    # squared L2 of the same pair is not representable: the batch reports it
    g2 = GpuIndex(nifs, 1)
    unwrap(nifs.flat_load_matrix(g2.ref, ids, x))
    assert nifs.flat_search_batch(g2.ref, qs, 5) == ("error", "metric overflow")
    with pytest.raises(oracle_mod.OracleError, match="metric overflow"):
        oracle_mod.matrix_search(1, x, packed, qs[0], 5)


def test_multi_query_scan_at_config_shape(nifs, oracle_mod):
    """d = 768 (three panels per row), N = 1M, a batch of 16 under manhattan (no GEMM form): two
    sweeps instead of sixteen scans, each query equal to its single search."""
    import torch
    from bench import build_shard, doc_ids
    rows, dim = 1_000_000, 768
    x = build_shard(torch, torch.device("cuda", 0), rows, dim, 77)
    g = GpuIndex(nifs, 5)
    assert nifs.flat_load_device_matrix(g.ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
    host = x[:4].cpu().numpy()
    del x
    qs = np.random.default_rng(1).uniform(-1, 1, (16, dim)).astype(np.float32)
    qs[:4] = host
    nifs.flat_set_profiling(g.ref, True)
    nifs.flat_get_profile(g.ref, reset=True)
    got = unwrap(nifs.flat_search_batch(g.ref, qs, 10))
    assert nifs.flat_get_profile(g.ref, reset=True)["scan_launches"] == 2
    for i in range(16):
        assert bits(got[i]) == bits(unwrap(nifs.flat_search(g.ref, qs[i], 10))), i
    for i in range(4):
        assert got[i][0] == (b"doc-%d" % (i + 1), 0.0)


@pytest.mark.parametrize("metric,d,n,limit", [(5, 100, 140_000, 10), (0, 300, 52_000, 10), (6, 320, 52_000, 7), (3, 72, 260_000, 20),
                                              (5, 256, 70_000, 100), (1, 768, 22_000, 256), (4, 64, 300_000, 33)])
def test_batches_as_k1p_sweeps_equal_single_queries(nifs, oracle_mod, metric, d, n, limit):
    """flat_search_batch on rows off K1m's 256-float panel grid, or with lists longer than K1m's wave buffers hold (32):
    groups of eight as K1p sweeps -- prefix_multi_kernel over the WHOLE row, `limit` candidates (host/vt_batch.h
    sweep_group_applies).  Every query's hits are its own flat_search's, bit for bit, and the oracle's."""
    x, ids = make_corpus(n, d, 9100 + metric + d, False, oracle_mod, tie_block=60)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    nifs.flat_set_profiling(g.ref, True)
    rng = np.random.default_rng(metric * 100 + d)
    packed = oracle_mod.pack_ids(ids)
    for nq in (2, 8, 11):
        qs = rng.uniform(-1, 1, size=(nq, d)).astype(np.float32)
        qs[0] = x[n // 2]                  # sits on the block of identical rows
        nifs.flat_get_profile(g.ref, reset=True)
        got = unwrap(nifs.flat_search_batch(g.ref, qs, limit))
        prof = nifs.flat_get_profile(g.ref, reset=True)
        assert prof["sweep_queries"] >= nq - 1 - nq // 8, (nq, prof)     # (a lone last query goes alone; a threshold may miss)
        for i in range(nq):
            assert bits(got[i]) == bits(unwrap(nifs.flat_search(g.ref, qs[i], limit))), (metric, d, nq, i)
        for i in (0, nq - 1):
            want = oracle_mod.matrix_search(metric, x, packed, qs[i], limit)
            assert bits(got[i]) == bits(want), (metric, d, nq, i)


@pytest.mark.parametrize("metric", [2, 0, 5, 3])
def test_many_groups_of_one_call_alternate_between_two_contexts(nifs, oracle_mod, vt_debug, metric):
    """Round 5 (host/vt_funnel.h funnel_groups): the groups of eight of ONE funnel_search_batch / flat_search_batch call
    alternate between the caller's context and a second one -- group g + 1 is queued before group g is waited for.  43
    queries = five groups and a straggler (plus a lone one that goes alone).  Every list equals the query's single call,
    the oracle's composition for a few, and the same call with the groups in series (`no_group_pipeline`); an overflow
    inside the MIDDLE group sends that group's queries on one by one and the call reports what the loop of single
    calls would."""
    n, d, nq = 30_000, 136, 43
    x, ids = make_corpus(n, d, 5900 + metric, metric == 2, oracle_mod, tie_block=200)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rows = [(ids[i], x[i]) for i in range(n)]
    by_id = dict(rows)
    rng = np.random.default_rng(77 + metric)
    qs = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    qs[0], qs[20] = x[n // 2], x[n // 2 + 3]                 # inside the block of identical rows, in two groups
    if metric == 2:
        qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
    stages, cand, limit = [40, 96], 60, 10
    nifs.flat_set_profiling(g.ref, True)
    nifs.flat_get_profile(g.ref, reset=True)
    got = unwrap(nifs.flat_funnel_search_batch(g.ref, qs, stages, cand, limit))
    prof = nifs.flat_get_profile(g.ref, reset=True)
    assert prof["prefix_queries"] >= nq - 1 - nq // 8, prof       # the sweeps took them (a threshold may miss now and then)
    nifs.flat_set_profiling(g.ref, False)
    for i in range(nq):
        assert bits(got[i]) == bits(unwrap(nifs.flat_funnel_search(g.ref, qs[i], stages, cand, limit))), (metric, i)
    for i in (0, 20, nq - 1):
        cur = rows
        for st in stages:
            cur = [(j, by_id[j]) for j, _ in oracle_mod.vector_top_k(cur, qs[i], metric, st, cand)]
        assert bits(got[i]) == bits(oracle_mod.vector_top_k(cur, qs[i], metric, d, limit)), (metric, i)
    vt_debug.set("no_group_pipeline", 1)
    series = unwrap(nifs.flat_funnel_search_batch(g.ref, qs, stages, cand, limit))
    vt_debug.set("no_group_pipeline", 0)
    assert [bits(h) for h in series] == [bits(h) for h in got]
    # plain searches of a metric without a GEMM form travel as whole-row sweeps of eight through the same groups
    if metric == 5:
        vt_debug.set("force_sweep_groups", 1)
        nifs.flat_set_profiling(g.ref, True)
        nifs.flat_get_profile(g.ref, reset=True)
        plain = unwrap(nifs.flat_search_batch(g.ref, qs, limit))
        assert nifs.flat_get_profile(g.ref, reset=True)["sweep_queries"] >= nq - 1 - nq // 8
        packed = oracle_mod.pack_ids(ids)
        for i in range(nq):
            assert bits(plain[i]) == bits(unwrap(nifs.flat_search(g.ref, qs[i], limit))), i
        for i in (0, 20, nq - 1):
            assert bits(plain[i]) == bits(oracle_mod.matrix_search(metric, x, packed, qs[i], limit)), i
    # an overflow in the third group of five (inner product): one query's own, then a stored row's (every group's)
    if metric == 3:
        bad = qs.copy()
        bad[19, :8] = 3e38
        singles = [nifs.flat_funnel_search(g.ref, q, [16], 50, 5) for q in bad]
        res = nifs.flat_funnel_search_batch(g.ref, bad, [16], 50, 5)
        firsterr = next((r for r in singles if r[0] == "error"), None)
        assert firsterr is not None and res == firsterr
        ok_part = unwrap(nifs.flat_funnel_search_batch(g.ref, np.delete(bad, 19, axis=0), [16], 50, 5))
        assert [bits(h) for h in ok_part] == [bits(unwrap(r)) for i, r in enumerate(singles) if i != 19]
        big = GpuIndex(nifs, 3)
        xb = x[:20_000].copy()
        xb[77, :8] = 3e38
        unwrap(nifs.flat_load_matrix(big.ref, ids[:20_000], xb))
        singles = [nifs.flat_funnel_search(big.ref, q, [16], 50, 5) for q in bad[:30]]
        res = nifs.flat_funnel_search_batch(big.ref, bad[:30], [16], 50, 5)
        firsterr = next((r for r in singles if r[0] == "error"), None)
        if firsterr is not None:
            assert res == firsterr
        else:
            assert [bits(h) for h in unwrap(res)] == [bits(unwrap(r)) for r in singles]


@pytest.mark.parametrize("metric", [2, 0])
def test_quantized_batches_of_many_groups_on_two_streams(nifs, oracle_mod, vt_debug, metric):
    """quantized_search_batch with several groups of eight (profiling off: the timed form waits group by group): every
    group is queued before anything is waited for, even groups on the caller's context, odd groups on a second one
    (round 5; `no_group_pipeline=1` waits group by group: the form profiled calls take).  Both forms
    give every query its own quantized_search's hits, bit for bit, and the oracle's composition (binary_top_k, then
    vector_top_k over the candidates: collection.ex:276-295)."""
    n, d, nq, cand, limit = 40_000, 200, 45, 80, 10
    x, ids = make_corpus(n, d, 6400 + metric, metric == 2, oracle_mod, tie_block=150)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(5 + metric)
    qs = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    qs[0], qs[9], qs[44] = x[n // 2], x[n // 2 + 5], x[n // 2]        # the block of identical rows, from three groups
    if metric == 2:
        qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
    singles = [bits(unwrap(nifs.flat_quantized_search(g.ref, q, cand, limit))) for q in qs]
    for name, value in (("no_group_pipeline", 0), ("no_group_pipeline", 1)):
        vt_debug.set(name, value)
        for _ in range(2):      # (the second call reuses both contexts' slots)
            got = unwrap(nifs.flat_quantized_search_batch(g.ref, qs, cand, limit))
            assert [bits(h) for h in got] == singles, (metric, name, value)
        vt_debug.set(name, 0)
    sign = x >= 0
    for i in (0, 9, 44):
        ham = (sign != (qs[i] >= 0)[None, :]).sum(axis=1)
        order = sorted(range(n), key=lambda r: (int(ham[r]), ids[r]))[:cand]
        want = oracle_mod.vector_top_k([(ids[r], x[r]) for r in order], qs[i], metric, d, limit)
        assert singles[i] == bits(want), (metric, i)


@pytest.mark.parametrize("metric", [2, 0, 5])
def test_funnel_groups_take_their_thresholds_from_tile_maxima_on_larger_corpora(nifs, oracle_mod, vt_debug, metric):
    """Round 5: where the sample is a fraction of the corpus (here 200 000 rows: every fourth 64-row tile) the sample pass
    of a funnel group files the best score of each tile and the threshold is the rank-th largest maximum
    (launch_sample_tau_groups) instead of a radix select over every sampled score (the form small corpora keep).
    A threshold only decides which rows are LISTED: every query must get its own funnel_search's hits and the oracle's
    composition, bit for bit -- rows identical to the query included."""
    n, d, nq = 200_000, 64, 21
    x, ids = make_corpus(n, d, 7300 + metric, metric == 2, oracle_mod, tie_block=40)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(17 + metric)
    qs = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    qs[0], qs[12] = x[n // 2], x[77]
    if metric == 2:
        qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
    stages, cand, limit = [24], 10, 5
    nifs.flat_set_profiling(g.ref, True)
    nifs.flat_get_profile(g.ref, reset=True)
    got = unwrap(nifs.flat_funnel_search_batch(g.ref, qs, stages, cand, limit))
    prof = nifs.flat_get_profile(g.ref, reset=True)
    nifs.flat_set_profiling(g.ref, False)
    assert prof["prefix_queries"] >= nq - 1 - nq // 8 - 2, prof      # the sweeps took them (a threshold may miss now and then)
    singles = [bits(unwrap(nifs.flat_funnel_search(g.ref, q, stages, cand, limit))) for q in qs]
    assert [bits(h) for h in got] == singles
    rows = [(ids[i], x[i]) for i in range(n)]
    by_id = dict(rows)
    for i in (0, 12):
        cur = [(j, by_id[j]) for j, _ in oracle_mod.vector_top_k(rows, qs[i], metric, stages[0], cand)]
        assert singles[i] == bits(oracle_mod.vector_top_k(cur, qs[i], metric, d, limit)), (metric, i)


@pytest.mark.parametrize("metric", [2, 0])
def test_funnel_group_lists_of_every_length_are_cut_alike(nifs, oracle_mod, metric):
    """The list select of a funnel group (round 5): lists of up to 2 048 keys are cut on sixteen blocks per list
    (select_lists_spread_kernel), longer ones by the one-block radix form beside it, lists beyond the cap send their
    query down the single path.  Three thousand rows identical to the query put 3 000 equal scores into its list (only
    the id ranks order them), six thousand two hundred overflow nothing but fill most of the cap; their neighbours in the
    same group have short lists.  Every query: its own funnel_search's hits and the oracle's."""
    n, d = 40_000, 72
    x, ids = make_corpus(n, d, 8800 + metric, metric == 2, oracle_mod)
    x[5000:8000] = x[5000]          # 3 000 equal rows
    x[20000:26200] = x[20000]       # 6 200 equal rows
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(3 + metric)
    qs = rng.uniform(-1, 1, (13, d)).astype(np.float32)
    if metric == 2:
        qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
    qs[1], qs[6], qs[9] = x[5000], x[20000], x[5001]
    stages, cand, limit = [32], 70, 12
    got = unwrap(nifs.flat_funnel_search_batch(g.ref, qs, stages, cand, limit))
    rows = [(ids[i], x[i]) for i in range(n)]
    by_id = dict(rows)
    for i in range(13):
        assert bits(got[i]) == bits(unwrap(nifs.flat_funnel_search(g.ref, qs[i], stages, cand, limit))), (metric, i)
    for i in (1, 6, 9, 12):
        cur = [(j, by_id[j]) for j, _ in oracle_mod.vector_top_k(rows, qs[i], metric, stages[0], cand)]
        assert bits(got[i]) == bits(oracle_mod.vector_top_k(cur, qs[i], metric, d, limit)), (metric, i)
    # the duplicates come back in id order, the first of them first
    assert [h[0] for h in got[1]] == sorted(ids[5000:8000])[:limit]
