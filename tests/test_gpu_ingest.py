"""Bulk loads (vt_flat_load_matrix of >= 65 536 rows on a one-shard handle) run their phases side by
side -- rows to the device, finiteness check, id table, id ranks (host/vt_store.h
index_store_bulk_host) -- and must still be flat.rs:69-85: the whole batch validated before anything
is stored, the last duplicate of an id wins, ids in any order."""
import numpy as np
import pytest

import support
from test_gpu_parity import GpuIndex, bits, nifs, unwrap  # noqa: F401  (nifs: fixture)

pytestmark = pytest.mark.gpu


def corpus(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    x[n // 3:n // 3 + 40] = x[n // 3]   # identical rows: the id bytes decide
    return x


def check(nifs, oracle_mod, g, metric, x, ids, seed, k=12):
    packed = oracle_mod.pack_ids(ids)
    rng = np.random.default_rng(seed)
    for q in (x[len(x) // 3], rng.uniform(-1, 1, x.shape[1]).astype(np.float32), x[-1]):
        assert bits(g.search(q, k)) == bits(oracle_mod.matrix_search(metric, x, packed, q, k))


@pytest.mark.parametrize("chunk_mb", [None, 2])
@pytest.mark.parametrize("order", ["sorted", "unsorted"])
def test_bulk_loads_in_both_id_orders(nifs, oracle_mod, order, chunk_mb, monkeypatch, vt_debug):
    """chunk_mb = 2: the slab is a mapped range of 2-MiB chunks (1 GiB in production), so these small loads map
    their chunks on a thread AHEAD of the copy, as a 30-GB load does."""
    if chunk_mb:
        vt_debug.set("slab_chunk_mb", chunk_mb)
    n, d = 150_000, 24
    x = corpus(n, d, 1)
    ids = [b"doc-%07d" % i for i in range(n)] if order == "sorted" else [b"doc-%d" % (i * 7919 % n) for i in range(n)]
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    assert len(g) == n and g.dimension == d
    check(nifs, oracle_mod, g, 0, x, ids, 2)
    # a second bulk load behind the first (appended rows; the ids interleave with the old ones)
    m = 90_000
    x2 = corpus(m, d, 3)
    ids2 = [b"doc-%07d-b" % i for i in range(m)] if order == "sorted" else [b"e%d" % (i * 104729 % m) for i in range(m)]
    unwrap(nifs.flat_load_matrix(g.ref, ids2, x2))
    check(nifs, oracle_mod, g, 0, np.concatenate([x, x2]), ids + ids2, 4)
    # batches of queries need strictly current ranks
    allx, allids = np.concatenate([x, x2]), ids + ids2
    qs = np.random.default_rng(5).uniform(-1, 1, size=(9, d)).astype(np.float32)
    got = unwrap(nifs.flat_search_batch(g.ref, qs, 5))
    packed = oracle_mod.pack_ids(allids)
    for i in range(9):
        assert bits(got[i]) == bits(oracle_mod.matrix_search(0, allx, packed, qs[i], 5))


@pytest.mark.parametrize("chunk_mb", [None, 2])
def test_a_non_finite_row_rejects_the_whole_bulk_load(nifs, oracle_mod, chunk_mb, monkeypatch, vt_debug):
    if chunk_mb:
        vt_debug.set("slab_chunk_mb", chunk_mb)
    n, d = 120_000, 16
    x = corpus(n, d, 7)
    ids = [b"r%d" % i for i in range(n)]
    g = GpuIndex(nifs, 2)
    bad = x.copy()
    bad[n - 5, 3] = np.inf   # near the end: most rows are on their way to the device by the time it is found
    assert nifs.flat_load_matrix(g.ref, ids, bad) == ("error", "vector contains a non-finite value")
    assert len(g) == 0 and g.dimension is None
    # the index takes another dimension afterwards (nothing of the rejected batch stayed behind)
    x8 = corpus(70_000, 8, 8)
    ids8 = [b"s%d" % i for i in range(len(x8))]
    unwrap(nifs.flat_load_matrix(g.ref, ids8, x8))
    check(nifs, oracle_mod, g, 2, x8, ids8, 9)
    # rejected on top of a loaded index: the rows that were there answer as before, and the free rows behind them are zeros again
    bad8 = corpus(80_000, 8, 10)
    bad8[100, 0] = np.nan
    assert nifs.flat_load_matrix(g.ref, [b"t%d" % i for i in range(len(bad8))], bad8) == ("error", "vector contains a non-finite value")
    assert len(g) == len(x8)
    check(nifs, oracle_mod, g, 2, x8, ids8, 11)
    more = corpus(66_000, 8, 12)
    ids_more = [b"u%d" % i for i in range(len(more))]
    unwrap(nifs.flat_load_matrix(g.ref, ids_more, more))
    check(nifs, oracle_mod, g, 2, np.concatenate([x8, more]), ids8 + ids_more, 13)


def test_upserts_and_duplicates_inside_a_bulk_load(nifs, oracle_mod):
    """Not every id new and distinct: the batch takes the general path after all (flat.rs:270-281: last wins)."""
    n, d = 100_000, 16
    x = corpus(n, d, 21)
    ids = [b"k%d" % i for i in range(n)]
    g = GpuIndex(nifs, 1)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    x2 = corpus(80_000, d, 22)
    ids2 = [b"k%d" % (i * 3) if i % 4 == 0 and i * 3 < n else b"n%d" % i for i in range(len(x2))]   # a quarter upserts
    ids2[70_001] = ids2[70_000]          # ... and a duplicate inside the batch
    unwrap(nifs.flat_load_matrix(g.ref, ids2, x2))
    want = oracle_mod.FlatIndex(1)
    want.insert_matrix(ids, x)
    want.insert_matrix(ids2, x2)
    assert len(g) == len(want)
    rng = np.random.default_rng(23)
    for q in (x2[70_001], x2[0], rng.uniform(-1, 1, d).astype(np.float32)):
        assert bits(g.search(q, 10)) == bits(want.search(q, 10))


@pytest.mark.parametrize("start", ["empty", "sorted corpus"])
@pytest.mark.parametrize("case", ["one duplicate", "upserts and newcomers"])
def test_sorted_ids_that_are_not_all_new_keep_the_rank_column_current(nifs, oracle_mod, start, case):
    """a bulk load whose ids are ascending (the id ranks stay valid, extended in place) but not all new and
    distinct -- one duplicate, or a sorted snapshot reloaded over the ids that are there plus new ones -- hands over to
    the general path AFTER every id is in the table.  The device's rank column must still receive the new rows' ranks
    (it was left stale, or null on a first load): identical rows, which only the id order tells apart, single searches
    and a batch, on an empty index and on one whose ranks are clean."""
    d = 16
    width = 7
    sid = lambda i: b"s%0*d" % (width, i)          # bytewise order == numeric order
    g = GpuIndex(nifs, 1)
    want = oracle_mod.FlatIndex(1)
    n0 = 0
    if start == "sorted corpus":
        n0 = 70_000
        x0 = corpus(n0, d, 61)
        ids0 = [sid(2 * i) for i in range(n0)]     # even numbers
        unwrap(nifs.flat_load_matrix(g.ref, ids0, x0))
        want.insert_matrix(ids0, x0)
    n = 80_000
    x = corpus(n, d, 62)
    x[5000:5040] = x[100]                            # rows only the id bytes order
    if case == "one duplicate":
        ids = [sid(2 * n0 + 10 + i) for i in range(n)]     # all above what is there, ascending ...
        ids[40_001] = ids[40_000]                          # ... with one id twice (the last one wins, flat.rs:270-281)
    else:
        # ascending ids: 30 000 that are in the index already (upserts, a sorted snapshot loaded again), then new ones above
        ids = [sid(2 * i) for i in range(0, 60_000, 2)] + [sid(2 * n0 + 10 + j) for j in range(n - 30_000)]
        if start == "empty":
            ids = [sid(10 + i) for i in range(n)]
            ids[123] = ids[122]
            ids[70_000] = ids[69_999]
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    want.insert_matrix(ids, x)
    assert len(g) == len(want)
    rng = np.random.default_rng(63)
    qs = np.stack([x[100], x[5010], x[40_001], x[n - 1], rng.uniform(-1, 1, d).astype(np.float32)])
    for q in qs:
        assert bits(g.search(q, 50)) == bits(want.search(q, 50))
    got = unwrap(nifs.flat_search_batch(g.ref, qs, 50))
    for i, q in enumerate(qs):
        assert bits(got[i]) == bits(want.search(q, 50)), i
    # and the index goes on: a sorted append behind it, another search
    more = corpus(70_000, d, 64)
    ids_more = [sid(9_000_000 + i) for i in range(len(more))]
    unwrap(nifs.flat_load_matrix(g.ref, ids_more, more))
    want.insert_matrix(ids_more, more)
    for q in (x[5010], more[len(more) // 3 + 3]):
        assert bits(g.search(q, 50)) == bits(want.search(q, 50))


def test_the_serial_path_gives_the_same_index(nifs, oracle_mod):
    """Loads of fewer than 65 536 rows run their phases one after the other (check, room, ids, rows: the r03 path, which
    large loads left in r04): the same rows in three such pieces give the index one pipelined load gives."""
    n, d = 70_000, 16
    x = corpus(n, d, 31)
    ids = [b"z%d" % (i * 31 % n) for i in range(n)]
    g = GpuIndex(nifs, 3)
    for lo, hi in ((0, 30_000), (30_000, 30_001), (30_001, n)):
        unwrap(nifs.flat_load_matrix(g.ref, ids[lo:hi], x[lo:hi]))
    check(nifs, oracle_mod, g, 3, x, ids, 32)
    one = GpuIndex(nifs, 3)
    unwrap(nifs.flat_load_matrix(one.ref, ids, x))
    for q in (x[5], x[n - 1], x[30_000]):
        assert bits(g.search(q, 40)) == bits(one.search(q, 40))


def test_ids_that_went_in_before_a_bad_row_was_found_come_out_again(nifs, oracle_mod, monkeypatch, vt_debug):
    """The check rides on the copy (host/vt_store.h: the threads that fill the pinned quarters look at the rows they
    copy), and the id thread follows the verified mark -- so a non-finite row near the END of a batch is found when
    most ids are in the table already.  flat.rs:69-85: nothing of the batch may stay -- ids, ranks (in-place for
    ascending ids, lazy otherwise), the dimension of an empty index, the rows behind the index.  Quarters of 1 MiB
    (VT_INGEST_STAGE_MB: 4 096 rows of this width) make the batches cross twenty of them."""
    vt_debug.set("ingest_stage_mb", 1)
    n, d = 90_000, 16
    x = corpus(n, d, 41)
    g = GpuIndex(nifs, 0)
    # (1) on an empty index, ascending ids (their ranks are written as they arrive)
    bad = x.copy()
    bad[n - 3, 5] = np.nan
    ids = [b"a%07d" % i for i in range(n)]
    assert nifs.flat_load_matrix(g.ref, ids, bad) == ("error", "vector contains a non-finite value")
    assert len(g) == 0 and g.dimension is None
    assert unwrap(nifs.flat_search(g.ref, [1.0, 2.0], 3)) == []          # (any dimension: the index is empty)
    # (2) the same ids, clean, are all new afterwards
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    check(nifs, oracle_mod, g, 0, x, ids, 42)
    # (3) on top of it, ids out of order (lazy ranks), the bad row in the last quarter: the old rows answer as before,
    # singly and in a batch (strict ranks), none of the new ids exists
    more = corpus(80_000, d, 43)
    ids_more = [b"m%d" % (i * 7919 % 80_000) for i in range(80_000)]
    bad2 = more.copy()
    bad2[79_990, 0] = -np.inf
    assert nifs.flat_load_matrix(g.ref, ids_more, bad2) == ("error", "vector contains a non-finite value")
    assert len(g) == n
    check(nifs, oracle_mod, g, 0, x, ids, 44)
    qs = np.random.default_rng(45).uniform(-1, 1, size=(5, d)).astype(np.float32)
    packed = oracle_mod.pack_ids(ids)
    got = unwrap(nifs.flat_search_batch(g.ref, qs, 7))
    for i in range(5):
        assert bits(got[i]) == bits(oracle_mod.matrix_search(0, x, packed, qs[i], 7))
    unwrap(nifs.flat_delete(g.ref, ids_more[5]))                         # (a no-op: the id never made it)
    assert len(g) == n
    # (4) ... and the clean batch goes in
    unwrap(nifs.flat_load_matrix(g.ref, ids_more, more))
    check(nifs, oracle_mod, g, 0, np.concatenate([x, more]), ids + ids_more, 46)


def test_every_id_placed_before_the_bad_row_is_taken_back(nifs, oracle_mod, request, monkeypatch, capfd, vt_debug):
    """The same with the race taken out (VT_TEST_INGEST_LOCKSTEP, libvettore_hip_hooks.so only: the test re-runs itself
    there): the id thread keeps step with the verified quarters, so when the bad row of the LAST quarter is found every
    id of the earlier ones is in the table -- and the trace says how many came out again."""
    if support.rerun_with_hooks_library(request):
        return
    vt_debug.set("ingest_stage_mb", 1)
    vt_debug.set("test_ingest_lockstep", 1)
    vt_debug.set("trace_ingest", 1)
    n, d = 90_000, 16                                    # rows are 256 B in the slab: 1-MiB quarters of 4 096 rows
    x = corpus(n, d, 51)
    ids = [b"a%07d" % i for i in range(n)]
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    more = corpus(80_000, d, 52)
    ids_more = [b"m%d" % (i * 7919 % 80_000) for i in range(80_000)]
    bad = more.copy()
    bad[79_999, 15] = np.inf
    capfd.readouterr()
    assert nifs.flat_load_matrix(g.ref, ids_more, bad) == ("error", "vector contains a non-finite value")
    err = capfd.readouterr().err
    assert "77824 ids taken back" in err, err            # nineteen whole quarters were verified (and their ids placed) before the twentieth failed
    assert len(g) == n
    check(nifs, oracle_mod, g, 0, x, ids, 53)
    unwrap(nifs.flat_load_matrix(g.ref, ids_more, more))
    check(nifs, oracle_mod, g, 0, np.concatenate([x, more]), ids + ids_more, 54)


def test_a_row_is_seen_by_every_reader_the_moment_its_insert_has_returned(nifs, oracle_mod):
    """One-row inserts, upserts and deletes return with their copies QUEUED (round 6: the landing ring, host/vt_types.h):
    readers on other contexts make their streams wait for the newest mutation's event.  So, round after round: a row is
    upserted to equal a query (or inserted under a new id, or the previous best hit deleted), and the moment the call is
    back four threads search -- each on a context of its own, the primary one and three leased ones -- and every one of
    them must already see it: hits equal to the oracle's over the rows as they now are, bit for bit.  flat.rs:59-93 under
    nifs.rs:259-309's lock: a write that has returned is visible to every later read."""
    import threading
    n, d, metric = 6000, 96, 3
    rng = np.random.default_rng(31)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    ids = [b"doc-%05d" % i for i in range(n)]
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    want = oracle_mod.FlatIndex(metric)
    want.insert_matrix(ids, x)
    barrier = threading.Barrier(5)
    state = {"q": None, "stop": False, "bad": []}

    def reader(t):
        while True:
            barrier.wait()                       # the writer's call has returned
            if state["stop"]:
                return
            got = unwrap(nifs.flat_search(g.ref, state["q"], 5))
            if bits(got) != state["want"]:
                state["bad"].append((t, state["step"]))
            barrier.wait()                       # everybody has looked

    ths = [threading.Thread(target=reader, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    try:
        for step in range(120):
            q = (rng.uniform(-1, 1, d) * 3.0).astype(np.float32)
            kind = step % 3
            if kind == 0:      # upsert an existing row to be the query's best hit
                key = ids[int(rng.integers(0, n))]
                unwrap(nifs.flat_insert(g.ref, key, q))
                want.insert(key, q)
            elif kind == 1:    # a new id
                key = b"new-%04d" % step
                unwrap(nifs.flat_insert(g.ref, key, q))
                want.insert(key, q)
            else:              # delete the row the last round made the best hit (the last row moves into its place)
                unwrap(nifs.flat_delete(g.ref, last_key))
                want.delete(last_key)
                q = last_q
            last_key, last_q = key, q
            state.update(q=q, want=bits(want.search(q, 5)), step=step)
            barrier.wait()
            barrier.wait()
    finally:
        state["stop"] = True
        barrier.wait()
        for th in ths:
            th.join()
    assert not state["bad"], state["bad"][:5]
    assert len(g) == len(want)


@pytest.mark.parametrize("d", [8, 100, 768])
def test_trickles_of_one_to_a_slotful_of_rows_land_as_the_oracle_says(nifs, oracle_mod, d):
    """Batches small enough for a slot of the landing ring (64 KiB: 21 rows of 768 floats, 126 of 100, 248 of 8) go down in ONE
    kernel launch -- rows, their slab rows, the ranks of ascending new ids -- and the call returns before it has run (round 6).
    Every shape of such a batch against the oracle index, searched right behind it: one row, a slotful, one row more than
    a slot holds (the staged path), ascending new ids (ranks ride along), ids out of order (lazy ranks), upserts mixed
    with new ids, an id twice in one batch (the LAST occurrence is the row, flat.rs:270-281) and three times, deletes in
    between (the last row moves into the hole in one launch)."""
    metric = 0
    ld = (d + 63) // 64 * 64
    slotful = (64 << 10) // (ld * 4 + 8)
    rng = np.random.default_rng(700 + d)
    g = GpuIndex(nifs, metric)
    want = oracle_mod.FlatIndex(metric)
    serial = [0]

    def vec():
        return rng.uniform(-1, 1, d).astype(np.float32)

    def both(items):
        g.insert_many(items)
        want.insert_many(items)
        assert len(g) == len(want)
        for q in (items[-1][1], vec()):
            assert bits(g.search(q, 12)) == bits(want.search(q, 12))

    def fresh(n, ascending=True):
        keys = []
        for _ in range(n):
            serial[0] += 1
            keys.append("k-%08d" % serial[0] if ascending else "u-%d" % ((serial[0] * 7919) % 100003))
        return keys

    both([(k, vec()) for k in fresh(1)])                                   # one row into an empty index
    both([(k, vec()) for k in fresh(slotful)])                             # a slotful of ascending new ids
    both([(k, vec()) for k in fresh(slotful + 1)])                         # one more than a slot holds: the staged path
    both([(k, vec()) for k in fresh(5, ascending=False)])                  # out of order: ranks go lazy
    both([(k, vec()) for k in fresh(3)])                                   # ascending again, on lazy ranks
    live = ["k-%08d" % i for i in range(1, 6)]
    both([(live[0], vec()), ("k-%08d" % (serial[0] + 1), vec()), (live[3], vec())])   # upserts around a new id
    serial[0] += 1
    twice = vec()
    both([("dup", vec()), ("k-%08d" % 2, vec()), ("dup", twice)])          # an id twice: the last occurrence is the row
    assert bits(g.search(twice, 1)) == bits(want.search(twice, 1)) and g.search(twice, 1)[0][0] == b"dup"
    thrice = vec()
    both([("tri", vec()), ("tri", vec()), ("tri", thrice)])
    assert g.search(thrice, 1)[0][0] == b"tri"
    for victim in ("dup", live[1], "tri"):                                 # deletes: swap with the last row, one launch
        g.delete(victim)
        want.delete(victim)
        q = vec()
        assert len(g) == len(want) and bits(g.search(q, 12)) == bits(want.search(q, 12))
    both([(k, vec()) for k in fresh(min(slotful, 9))])                     # and on it goes
    for _ in range(30):                                                    # a run of single puts and deletes, searched each time
        if rng.integers(0, 3) == 0 and len(want) > 3:
            victim = "k-%08d" % int(rng.integers(1, serial[0] + 1))
            g.delete(victim)
            want.delete(victim)
        else:
            both([(fresh(1)[0] if rng.integers(0, 2) else "k-%08d" % int(rng.integers(1, serial[0] + 1)), vec())])
    q = vec()
    assert len(g) == len(want) and bits(g.search(q, 40)) == bits(want.search(q, 40))
