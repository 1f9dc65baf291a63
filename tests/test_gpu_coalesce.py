"""Searches that meet on one handle go together (coalesced_search, csrc/host/vt_coalesce.h): a search that
finds another one running waits, and what has queued up runs as one batch.  Every caller must
get exactly what its own search returns alone -- hits, order, raw bits, and its own error.
`VT_COALESCE_SLOTS=1` makes even a small corpus queue its callers."""
import threading

import numpy as np
import pytest

import support
from test_gpu_parity import GpuIndex, bits, make_corpus, nifs, unwrap  # noqa: F401  (nifs is a fixture)

pytestmark = pytest.mark.gpu


def _hammer(nifs, ref, jobs, threads, rounds):
    """jobs: [(query, limit, expected)], expected = bits list or ("error", text).  Returns the mismatches."""
    bad, lock = [], threading.Lock()
    start = threading.Barrier(threads)

    def worker(t):
        start.wait()
        for r in range(rounds):
            q, k, want = jobs[(t * 7 + r) % len(jobs)]
            res = nifs.flat_search(ref, q, k)
            got = bits(res[1]) if res[0] == "ok" else res
            if got != want:
                with lock:
                    bad.append((t, r, k, str(got)[:120], str(want)[:120]))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    return bad


@pytest.mark.parametrize("metric,devices", [(2, None), (0, None), (5, None), (2, [0, 0, 0])])
def test_coalesced_searches_equal_searches_alone(nifs, oracle_mod, monkeypatch, metric, devices, vt_debug):
    n, d = 40_000, 96
    x, ids = make_corpus(n, d, 910 + metric, metric == 2, oracle_mod, tie_block=40)
    g = GpuIndex(nifs, metric)
    if devices:
        g.ref = nifs.flat_new_sharded(metric, devices)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(5)
    qs = [x[n // 2], x[3]] + [rng.uniform(-1, 1, d).astype(np.float32) for _ in range(30)]
    if metric == 2:
        qs = [oracle_mod.normalize_l2(q) for q in qs]
    vt_debug.set("coalesce", 0)
    jobs = []
    for i, q in enumerate(qs):
        k = (10, 10, 10, 5, 32, 300)[i % 6]          # 300 > 256: never batched
        jobs.append((q, k, bits(unwrap(nifs.flat_search(g.ref, q, k)))))
    jobs.append((np.ones(d + 1, np.float32), 10, ("error", "dimension mismatch")))
    jobs.append((np.full(d, np.nan, np.float32), 10, ("error", "vector contains a non-finite value")))
    assert jobs[0][2] == bits(oracle_mod.matrix_search(metric, x, oracle_mod.pack_ids(ids), qs[0], 10))
    vt_debug.reset("coalesce")
    vt_debug.set("coalesce_slots", 1)
    before = nifs.flat_coalesce_stats(g.ref)
    bad = _hammer(nifs, g.ref, jobs, threads=16, rounds=60)
    after = nifs.flat_coalesce_stats(g.ref)
    assert not bad, bad[:3]
    # (whether any of these Python threads met is a matter of timing -- test_callers_made_to_meet_* below force it;
    # here: whoever met, everybody got the answer of a search alone, and a batch never carries fewer than two)
    assert after[1] - before[1] >= 2 * (after[0] - before[0])
    print("coalesced: %d batches, %d calls in batches" % (after[0] - before[0], after[1] - before[1]))


def test_coalescing_steps_aside_for_lazy_ranks_and_survives_mutations(nifs, oracle_mod, monkeypatch, vt_debug):
    """Unsorted inserts leave the id ranks lazy: a batch would force a re-rank that lone searches
    avoid, so the queued callers are released to search side by side; a writer keeps inserting
    rows that never reach a result while 12 readers check their answers."""
    n, d = 30_000, 64
    x, ids = make_corpus(n, d, 77, False, oracle_mod, tie_block=30)
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(9)
    qs = [x[n // 2]] + [rng.uniform(-1, 1, d).astype(np.float32) for _ in range(11)]
    jobs = [(q, 10, bits(unwrap(nifs.flat_search(g.ref, q, 10)))) for q in qs]
    far = (rng.uniform(-1, 1, (400, d)) * 0.01 + 50.0).astype(np.float32)
    vt_debug.set("coalesce_slots", 1)
    stop = threading.Event()

    def writer():
        i = 0
        while not stop.is_set():
            key = (b"aa-%d" if i % 2 else b"zz-%d") % (i % 400)
            assert nifs.flat_insert(g.ref, key, far[i % 400]) == ("ok", ())
            if i % 3 == 2:
                nifs.flat_delete(g.ref, key)
            i += 1

    w = threading.Thread(target=writer)
    w.start()
    try:
        bad = _hammer(nifs, g.ref, jobs, threads=12, rounds=80)
    finally:
        stop.set()
        w.join()
    assert not bad, bad[:3]


# ---- callers that are MADE to meet ------------------------------------------------------------------------------
# The tests above use Python threads and assert answers only.  These assert the PATH -- which batches formed, which
# kernels carried them -- and therefore take timing out of it: native threads leave a barrier together, and the handle's
# first caller of a round keeps its slot until all the others have queued behind it (test_coalesce_hold_until, a hook of
# libvettore_hip_hooks.so; each test re-runs itself in a process that loads that build).  What happens after they have
# met -- take_along, judge, the batch paths, waking the members -- is the product's code, unchanged.

@pytest.mark.parametrize("metric,devices", [(2, None), (0, None), (5, None), (7, None), (2, [0, 0, 0])])
def test_callers_made_to_meet_travel_as_one_batch(nifs, oracle_mod, request, vt_debug, metric, devices):
    """16 flat_search callers per round: one batch of 16 every round -- a matrix-core pass (cosine, L2), sweeps of eight
    (manhattan: K1m), sweeps of the non-zero-bit column (float hamming: K4p), a batch on every shard of a three-shard
    handle -- and every answer the oracle's."""
    if support.rerun_with_hooks_library(request):
        return
    vt_debug.set("coalesce_slots", 1)
    vt_debug.set("force_batch_mfma", 1)
    vt_debug.set("force_multi_scan", 1)
    n, d, callers, rounds = 40_000, 96, 16, 4
    x, ids = make_corpus(n, d, 910 + metric, metric == 2, oracle_mod, tie_block=40)
    if metric == 7:
        x = (x * (np.random.default_rng(3).uniform(0, 1, x.shape) < 0.4)).astype(np.float32)
    packed = oracle_mod.pack_ids(ids)
    g = GpuIndex(nifs, metric)
    if devices:
        g.ref = nifs.flat_new_sharded(metric, devices)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    nifs.flat_set_profiling(g.ref, True)
    rng = np.random.default_rng(5)
    qs = np.stack([x[n // 2], x[3]] + [rng.uniform(-1, 1, d).astype(np.float32) for _ in range(21)])
    if metric == 2:
        qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
    for q in qs:
        assert bits(unwrap(nifs.flat_search(g.ref, q, 10))) == bits(oracle_mod.matrix_search(metric, x, packed, q, 10))
    b0, p0 = nifs.flat_coalesce_stats(g.ref), nifs.flat_get_profile(g.ref)
    wrong, failed = support.callers_meet(g.ref, qs, 10, [0] * callers, rounds=rounds)
    b1, p1 = nifs.flat_coalesce_stats(g.ref), nifs.flat_get_profile(g.ref)
    assert (wrong, failed) == (0, 0)
    assert (b1[0] - b0[0], b1[1] - b0[1]) == (rounds, rounds * callers), (b0, b1)
    moved = {k: p1[k] - p0[k] for k in p1 if isinstance(p1[k], int)}
    if devices:
        return                                             # (profiles are per shard; the batch count above is the handle's)
    if metric in (2, 0):
        assert moved["nominate_launches"] >= rounds and moved["nominate_queries"] >= rounds * callers, moved
    elif metric == 5:
        # sweeps of eight (K1m), not sixteen scans a round: one launch per sweep -- on top of the rounds * callers searches
        # that callers_meet made ALONE first, to compare with
        assert moved["scan_launches"] == rounds * callers + rounds * (callers // 8), moved
    else:
        assert moved["hamming_queries"] == rounds * callers and moved["scan_launches"] == 0, moved


def test_callers_of_three_entry_points_made_to_meet_travel_with_their_equals(nifs, oracle_mod, request, vt_debug):
    """24 callers per round -- 8 flat_search, 6 quantized_search(100), 2 quantized_search(50), 5 funnel_search([32], 100),
    3 funnel_search([64], 100): they queue on one handle and leave in five batches, equals with equals
    (collection.ex:234-295 under one read lock); every answer is that of the call made alone."""
    if support.rerun_with_hooks_library(request):
        return
    vt_debug.set("coalesce_slots", 1)
    n, d, rounds = 40_000, 128, 3
    x, ids = make_corpus(n, d, 4200, True, oracle_mod)
    g = GpuIndex(nifs, 2)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    nifs.flat_set_profiling(g.ref, True)
    rng = np.random.default_rng(12)
    qs = np.stack([oracle_mod.normalize_l2(q) for q in rng.uniform(-1, 1, (32, d)).astype(np.float32)])
    kinds = [0] * 8 + [1] * 8 + [2] * 8
    params = [0] * 16 + [32] * 5 + [64] * 3
    cands = [0] * 8 + [100] * 6 + [50] * 2 + [100] * 8
    b0, p0 = nifs.flat_coalesce_stats(g.ref), nifs.flat_get_profile(g.ref)
    wrong, failed = support.callers_meet(g.ref, qs, 10, kinds, params, cands, rounds=rounds)
    b1, p1 = nifs.flat_coalesce_stats(g.ref), nifs.flat_get_profile(g.ref)
    assert (wrong, failed) == (0, 0)
    # five groups of equals per round; all 24 callers travelled in one of them
    assert (b1[0] - b0[0], b1[1] - b0[1]) == (5 * rounds, 24 * rounds), (b0, b1)
    assert p1["hamming_queries"] - p0["hamming_queries"] >= 8 * rounds, (p0, p1)      # the quantized groups shared sweeps of the bits
    assert p1["prefix_queries"] - p0["prefix_queries"] >= 6 * rounds, (p0, p1)        # the funnel groups shared stage-1 sweeps
