"""Searches that meet on one handle go together (coalesced_search, csrc/host/vt_coalesce.h): a search that
finds another one running waits, and what has queued up runs as one batch.  Every caller must
get exactly what its own search returns alone -- hits, order, raw bits, and its own error.
`VT_COALESCE_SLOTS=1` makes even a small corpus queue its callers."""
import threading

import numpy as np
import pytest

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
    assert after[0] > before[0] and after[1] - before[1] >= 2 * (after[0] - before[0])   # batches ran, >= 2 searches each


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
