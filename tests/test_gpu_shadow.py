"""GPU parity tests for K2s: query batches nominated from the bf16 SHADOW of the rows
(vettore_amd/csrc/vt_batch_shadow.hip).

The shadow is an accelerator of the nomination pass and must never show in a result: every
query of a batch equals its own flat_search and the oracle's restatement of flat.rs:96-124 bit
for bit -- with the shadow current, after mutations that leave it to be patched, with it
switched off, and when the card "has no room" for it.
"""
import numpy as np
import pytest

import support
from test_gpu_parity import GpuIndex, bits, make_corpus, nifs, unwrap  # noqa: F401  (nifs: fixture)

pytestmark = pytest.mark.gpu

SHADOW_OFF, SHADOW_AUTO = 0, 1


def check_batch(nifs, oracle_mod, g, metric, x, ids, qs, k, note=""):
    packed = oracle_mod.pack_ids(ids)
    got = unwrap(nifs.flat_search_batch(g.ref, qs, k))
    assert len(got) == len(qs)
    for i, q in enumerate(qs):
        assert bits(got[i]) == bits(oracle_mod.matrix_search(metric, x, packed, q, k)), (note, metric, len(qs), k, i)


@pytest.mark.parametrize("metric", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("d", [192, 100])
def test_batches_from_the_shadow_equal_the_oracle(nifs, oracle_mod, metric, d, monkeypatch, vt_debug):
    """All five matrix-core metrics, rows on and off the 64-float grid (d = 100 pads to 128), the
    three pass widths (64 / 128 / 256 query columns), a trailing group, limits 1..64."""
    vt_debug.set("force_batch_mfma", 1)   # the cost model would send these small corpora to single scans
    n = 20000
    x, ids = make_corpus(n, d, 4100 + metric, metric == 2, oracle_mod, tie_block=48)
    g = GpuIndex(nifs, metric)
    assert nifs.flat_batch_shadow(g.ref) == "none"
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(81)
    nifs.flat_set_profiling(g.ref, True)
    for nq, k in ((8, 10), (37, 1), (100, 10), (128, 3), (256, 10), (300, 64)):
        qs = rng.uniform(-1, 1, size=(nq, d)).astype(np.float32)
        qs[0] = x[n // 2]  # sits on the block of identical rows
        if metric == 2:
            qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
        check_batch(nifs, oracle_mod, g, metric, x, ids, qs, k)
    assert nifs.flat_batch_shadow(g.ref) == "current"
    prof = nifs.flat_get_profile(g.ref)
    assert prof["shadow_builds"] == 1, prof
    assert prof["nominate_launches"] >= 6 and prof["nominate_shadow_launches"] == prof["nominate_launches"], prof
    assert prof["batch_launches"] == 0, prof
    assert prof["batch_fallbacks"] <= prof["nominate_queries"] // 10, prof
    # the bytes the passes are priced at are the shadow's: rows * d * 2 each
    assert prof["nominate_bytes"] == prof["nominate_launches"] * n * d * 2, prof


def test_mutations_patch_the_shadow(nifs, oracle_mod, monkeypatch, vt_debug):
    """Upserts, deletes (swap with the last row), appends past the slab's capacity, emptying and
    re-dimensioning: the shadow follows (patched rows, rebuilds) and the batches keep equalling
    the oracle over the current rows."""
    vt_debug.set("force_batch_mfma", 1)
    n, d = 9000, 128
    metric = 3
    x, ids = make_corpus(n, d, 4242, False, oracle_mod, tie_block=20)
    x = x.copy()
    ids = list(ids)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    nifs.flat_set_profiling(g.ref, True)
    rng = np.random.default_rng(5)
    qs = rng.uniform(-1, 1, size=(40, d)).astype(np.float32)
    check_batch(nifs, oracle_mod, g, metric, x, ids, qs, 10, "fresh")
    assert nifs.flat_batch_shadow(g.ref) == "current"

    # upserts: the new rows must be the ones nominated (they are made the best hits of query 0)
    for r in (0, 17, n - 1, 4500):
        x[r] = (qs[0] * (3.0 + r % 5)).astype(np.float32)
        unwrap(nifs.flat_insert(g.ref, ids[r], x[r]))
    assert nifs.flat_batch_shadow(g.ref) == "stale"
    check_batch(nifs, oracle_mod, g, metric, x, ids, qs, 10, "upserts")
    assert nifs.flat_batch_shadow(g.ref) == "current"
    prof = nifs.flat_get_profile(g.ref)
    assert prof["shadow_builds"] == 1 and prof["shadow_patched_rows"] == 4, prof

    # deletes: the last row moves into the hole
    for r in (3, 4500, 100):
        last = len(ids) - 1
        unwrap(nifs.flat_delete(g.ref, ids[r]))
        if r != last:
            x[r] = x[last]
            ids[r] = ids[last]
        x = x[:last]
        ids.pop()
    check_batch(nifs, oracle_mod, g, metric, x, ids, qs, 10, "deletes")

    # appends in small and in bulk form (the second one outgrows the slab: a rebuild)
    extra = rng.uniform(-1, 1, size=(50, d)).astype(np.float32)
    extra[7] = (qs[1] * 5.0).astype(np.float32)
    new_ids = [b"new-%d" % i for i in range(50)]
    unwrap(nifs.flat_insert_many(g.ref, list(zip(new_ids, extra))))
    x = np.concatenate([x, extra])
    ids += new_ids
    check_batch(nifs, oracle_mod, g, metric, x, ids, qs, 10, "small append")
    bulk = rng.uniform(-1, 1, size=(70000, d)).astype(np.float32)
    bulk[69999] = (qs[2] * 7.0).astype(np.float32)
    bulk_ids = [b"bulk-%d" % i for i in range(len(bulk))]
    unwrap(nifs.flat_load_matrix(g.ref, bulk_ids, bulk))
    x = np.concatenate([x, bulk])
    ids += bulk_ids
    check_batch(nifs, oracle_mod, g, metric, x, ids, qs, 10, "bulk append")
    prof = nifs.flat_get_profile(g.ref)
    assert prof["shadow_builds"] >= 2, prof
    assert prof["nominate_shadow_launches"] == prof["nominate_launches"] >= 5, prof


def test_an_emptied_index_gives_the_shadow_back(nifs, oracle_mod, monkeypatch, vt_debug):
    vt_debug.set("force_batch_mfma", 1)
    n, d = 6000, 64
    x, ids = make_corpus(n, d, 77, False, oracle_mod)
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    qs = np.random.default_rng(1).uniform(-1, 1, size=(16, d)).astype(np.float32)
    check_batch(nifs, oracle_mod, g, 0, x, ids, qs, 5)
    assert nifs.flat_batch_shadow(g.ref) == "current"
    for i in ids:
        unwrap(nifs.flat_delete(g.ref, i))
    assert g.dimension is None
    # another dimension altogether
    d2 = 200
    x2, ids2 = make_corpus(5000, d2, 78, False, oracle_mod)
    unwrap(nifs.flat_load_matrix(g.ref, ids2, x2))
    assert nifs.flat_batch_shadow(g.ref) == "none"
    qs2 = np.random.default_rng(2).uniform(-1, 1, size=(70, d2)).astype(np.float32)
    check_batch(nifs, oracle_mod, g, 0, x2, ids2, qs2, 5)
    assert nifs.flat_batch_shadow(g.ref) == "current"


def test_the_shadow_switched_off_and_on(nifs, oracle_mod, monkeypatch, vt_debug):
    """VT_SHADOW_OFF: the pass streams the f32 rows (K2b) -- same hits; switching it back on builds anew."""
    vt_debug.set("force_batch_mfma", 1)
    n, d = 12000, 192
    x, ids = make_corpus(n, d, 9, True, oracle_mod, tie_block=30)
    g = GpuIndex(nifs, 2)
    assert nifs.flat_set_batch_shadow(g.ref, SHADOW_OFF) == ("ok", ())
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    nifs.flat_set_profiling(g.ref, True)
    qs = np.stack([oracle_mod.normalize_l2(q) for q in np.random.default_rng(3).uniform(-1, 1, size=(64, d)).astype(np.float32)])
    check_batch(nifs, oracle_mod, g, 2, x, ids, qs, 10, "off")
    assert nifs.flat_batch_shadow(g.ref) == "off"
    prof = nifs.flat_get_profile(g.ref, reset=True)
    assert prof["nominate_launches"] == 1 and prof["nominate_shadow_launches"] == 0 and prof["shadow_builds"] == 0, prof
    assert prof["nominate_bytes"] == n * d * 4, prof
    assert nifs.flat_set_batch_shadow(g.ref, SHADOW_AUTO) == ("ok", ())
    check_batch(nifs, oracle_mod, g, 2, x, ids, qs, 10, "on again")
    assert nifs.flat_batch_shadow(g.ref) == "current"
    prof = nifs.flat_get_profile(g.ref, reset=True)
    assert prof["nominate_shadow_launches"] == 1 and prof["shadow_builds"] == 1, prof
    assert nifs.flat_set_batch_shadow(g.ref, SHADOW_OFF) == ("ok", ())
    assert nifs.flat_batch_shadow(g.ref) == "off"
    check_batch(nifs, oracle_mod, g, 2, x, ids, qs, 10, "off again")
    # the f32 matrix cores never look at it
    assert nifs.flat_set_batch_shadow(g.ref, SHADOW_AUTO) == ("ok", ())
    assert nifs.flat_set_batch_nominate(g.ref, 1) == "ok"
    check_batch(nifs, oracle_mod, g, 2, x, ids, qs, 10, "f32 nomination")
    assert nifs.flat_batch_shadow(g.ref) == "none"


def test_no_room_for_the_shadow_means_streaming_the_rows(nifs, oracle_mod, request, monkeypatch, vt_debug):
    """The shadow is an accelerator: when the card has no room for it the batches keep reading the f32
    rows, with the same hits.  (The refused allocation is injected -- VT_TEST_REFUSE_SHADOW,
    libvettore_hip_hooks.so only: the test re-runs itself there.)"""
    if support.rerun_with_hooks_library(request):
        return
    vt_debug.set("test_refuse_shadow", 1)
    vt_debug.set("force_batch_mfma", 1)
    n, d = 10000, 128
    x, ids = make_corpus(n, d, 31, False, oracle_mod)
    g = GpuIndex(nifs, 1)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    nifs.flat_set_profiling(g.ref, True)
    qs = np.random.default_rng(4).uniform(-1, 1, size=(33, d)).astype(np.float32)
    for _ in range(2):
        check_batch(nifs, oracle_mod, g, 1, x, ids, qs, 7)
    assert nifs.flat_batch_shadow(g.ref) == "refused"
    prof = nifs.flat_get_profile(g.ref)
    assert prof["nominate_launches"] == 2 and prof["nominate_shadow_launches"] == 0 and prof["shadow_builds"] == 0, prof


def test_a_second_context_that_cannot_get_its_scratch_is_dropped(nifs, oracle_mod, request, vt_debug):
    """A batch call of several 256-query groups alternates between two contexts (round 5), which roughly doubles its device
    and pinned scratch.  When the second context's buffers do not fit -- a card nearly filled by corpus and shadow -- the
    call must go on in series on the first one, as it did before there was a second, not fail.
    (test_refuse_spare_scratch, libvettore_hip_hooks.so only: the test re-runs itself there.)"""
    if support.rerun_with_hooks_library(request):
        return
    vt_debug.set("force_batch_mfma", 1)
    n, d, metric = 20000, 128, 3
    x, ids = make_corpus(n, d, 515, False, oracle_mod, tie_block=30)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    nifs.flat_set_profiling(g.ref, False)
    qs = np.random.default_rng(8).uniform(-1, 1, size=(700, d)).astype(np.float32)
    qs[0], qs[300], qs[699] = x[n // 2], x[n // 2 + 3], x[n // 2]
    both = unwrap(nifs.flat_search_batch(g.ref, qs, 10))           # two contexts
    vt_debug.set("test_refuse_spare_scratch", 1)
    alone = unwrap(nifs.flat_search_batch(g.ref, qs, 10))          # the second one refused: in series on the first
    vt_debug.set("test_refuse_spare_scratch", 0)
    assert [bits(h) for h in alone] == [bits(h) for h in both]
    packed = oracle_mod.pack_ids(ids)
    for i in (0, 255, 256, 300, 511, 512, 699):
        assert bits(alone[i]) == bits(oracle_mod.matrix_search(metric, x, packed, qs[i], 10)), i


def test_rows_that_round_to_infinity_through_the_shadow(nifs, oracle_mod, monkeypatch, vt_debug):
    """f32's largest values round to +inf in bf16 -- in the shadow as in K2b's registers: inf * 0 = NaN
    nominates nothing, the handle's largest row norm keeps such a corpus from ever being certified,
    the exact paths answer."""
    vt_debug.set("force_batch_mfma", 1)
    n, d = 6000, 64
    rng = np.random.default_rng(22)
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    big = np.finfo(np.float32).max
    for r in (5, 700, 5999):
        x[r, 0] = big
        x[r, 1:] = 2.0
    ids = [b"r%d" % i for i in range(n)]
    g = GpuIndex(nifs, 3)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    qs = rng.uniform(-1, 1, size=(16, d)).astype(np.float32)
    qs[:, 0] = 0.0
    check_batch(nifs, oracle_mod, g, 3, x, ids, qs, 5)


def test_concurrent_callers_get_their_own_answers_beside_a_shadow(nifs, oracle_mod, monkeypatch, vt_debug):
    """flat_search callers on a handle whose batches go through the shadow: every answer is the oracle's, whoever
    travelled with whom.  (How many of these Python threads meet is a matter of timing and is not asserted: the test
    below makes callers meet.)"""
    import threading
    vt_debug.set("force_batch_mfma", 1)
    vt_debug.set("coalesce_slots", 1)
    n, d = 30000, 256
    x, ids = make_corpus(n, d, 61, True, oracle_mod, tie_block=16)
    packed = oracle_mod.pack_ids(ids)
    g = GpuIndex(nifs, 2)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(6)
    qs = np.stack([oracle_mod.normalize_l2(q) for q in rng.uniform(-1, 1, size=(24, d)).astype(np.float32)])
    want = [bits(oracle_mod.matrix_search(2, x, packed, q, 10)) for q in qs]
    errors = []
    start = threading.Barrier(len(qs))

    def caller(i):
        try:
            start.wait()
            for _ in range(6):
                got = unwrap(nifs.flat_search(g.ref, qs[i], 10))
                if bits(got) != want[i]:
                    errors.append(i)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=caller, args=(i,)) for i in range(len(qs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_callers_made_to_meet_travel_through_the_shadow(nifs, oracle_mod, request, vt_debug):
    """flat_search callers that meet on a handle go as one batch (vt_coalesce.h); with a shadow that batch is a K2s
    pass.  The meeting is FORCED: 24 native threads leave a barrier together and the handle's first caller keeps its
    slot until the other 23 have queued (test_coalesce_hold_until -- libvettore_hip_hooks.so only: the test re-runs
    itself there), so every round is one batch of 24 whatever the box's timing; a row upserted between two runs leaves
    the shadow stale and the next batch patches it."""
    if support.rerun_with_hooks_library(request):
        return
    vt_debug.set("force_batch_mfma", 1)
    vt_debug.set("coalesce_slots", 1)
    n, d, callers, rounds = 30000, 256, 24, 5
    x, ids = make_corpus(n, d, 61, True, oracle_mod, tie_block=16)
    x = x.copy()
    packed = oracle_mod.pack_ids(ids)
    g = GpuIndex(nifs, 2)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    nifs.flat_set_profiling(g.ref, True)
    rng = np.random.default_rng(6)
    qs = np.stack([oracle_mod.normalize_l2(q) for q in rng.uniform(-1, 1, size=(31, d)).astype(np.float32)])
    qs[0] = x[n // 2]                                      # on the block of identical rows
    for run in range(2):
        # alone, every query equals the oracle; callers_meet compares every answer of a batch with the call made alone
        for q in qs:
            assert bits(unwrap(nifs.flat_search(g.ref, q, 10))) == bits(oracle_mod.matrix_search(2, x, packed, q, 10))
        b0, p0 = nifs.flat_coalesce_stats(g.ref), nifs.flat_get_profile(g.ref)
        wrong, failed = support.callers_meet(g.ref, qs, 10, [0] * callers, rounds=rounds)
        b1, p1 = nifs.flat_coalesce_stats(g.ref), nifs.flat_get_profile(g.ref)
        assert (wrong, failed) == (0, 0)
        assert (b1[0] - b0[0], b1[1] - b0[1]) == (rounds, rounds * callers), (b0, b1)
        passes = p1["nominate_launches"] - p0["nominate_launches"]
        assert passes >= rounds and p1["nominate_shadow_launches"] - p0["nominate_shadow_launches"] == passes, (p0, p1)
        assert p1["nominate_queries"] - p0["nominate_queries"] >= rounds * callers, (p0, p1)
        assert p1["shadow_builds"] == 1, p1
        if run == 0:
            x[7] = qs[3]                                   # an upsert: the shadow is stale until the next batch patches the row
            unwrap(nifs.flat_insert(g.ref, ids[7], x[7]))
            assert nifs.flat_batch_shadow(g.ref) == "stale"
    assert nifs.flat_batch_shadow(g.ref) == "current" and p1["shadow_patched_rows"] >= 1, p1


def test_a_lone_search_through_the_shadow(nifs, oracle_mod):
    """vt_flat_set_single_nominate: flat_search as a batch of one -- the bf16 pass over the shadow nominates, the exact
    kernel decides; hits equal the oracle's (and the plain scan's) bit for bit, tie blocks included."""
    for metric in (2, 0, 3):
        n, d = 30000, 192
        x, ids = make_corpus(n, d, 900 + metric, metric == 2, oracle_mod, tie_block=40)
        packed = oracle_mod.pack_ids(ids)
        g = GpuIndex(nifs, metric)
        unwrap(nifs.flat_load_matrix(g.ref, ids, x))
        assert nifs.flat_set_single_nominate(g.ref, True) == ("ok", ())
        nifs.flat_set_profiling(g.ref, True)
        rng = np.random.default_rng(17)
        qs = [x[n // 2], x[7]] + [rng.uniform(-1, 1, d).astype(np.float32) for _ in range(6)]
        if metric == 2:
            qs = [oracle_mod.normalize_l2(q) for q in qs]
        for k in (1, 10, 100):
            for q in qs:
                assert bits(g.search(q, k)) == bits(oracle_mod.matrix_search(metric, x, packed, q, k)), (metric, k)
        assert bits(g.search(qs[2], 300)) == bits(oracle_mod.matrix_search(metric, x, packed, qs[2], 300))   # beyond one pass: the scan
        prof = nifs.flat_get_profile(g.ref)
        assert prof["nominate_shadow_launches"] >= 20 and prof["shadow_builds"] == 1, prof
        # after a mutation the next lone search patches the shadow and goes on
        x[5] = (qs[3] * 4).astype(np.float32) if metric != 2 else qs[3]
        unwrap(nifs.flat_insert(g.ref, ids[5], x[5]))
        assert bits(g.search(qs[3], 10)) == bits(oracle_mod.matrix_search(metric, x, packed, qs[3], 10))
