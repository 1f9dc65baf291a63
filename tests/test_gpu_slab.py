"""Where the rows live (`Slab`, csrc/host/vt_base.h): below one chunk a plain allocation regrown by
copy, from one chunk on a reserved virtual range into which equal physical chunks are mapped as
the rows arrive -- the rows never move, nothing is copied, and the slab is never larger than
the rows plus one chunk.  The tests shrink the chunk to 2 MiB (`VT_SLAB_CHUNK_MB`) so that a
few thousand rows cross many chunk borders; the full-size tests run the real 1-GiB chunks
(30 of them under the 10 M-row corpus)."""
import numpy as np
import pytest

import support

from test_gpu_parity import GpuIndex, bits, nifs, unwrap  # noqa: F401  (nifs is a fixture)

pytestmark = pytest.mark.gpu

CHUNK = 2 << 20


def _check(g, o, qs, limits=(1, 10, 40)):
    for q in qs:
        for k in limits:
            assert bits(g.search(q, k)) == bits(o.search(q, k))


@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_slab_grows_by_mapping_chunks(nifs, oracle_mod, monkeypatch, devices, vt_debug):
    vt_debug.set("slab_chunk_mb", 2)
    d, row_bytes = 64, 256
    rng = np.random.default_rng(21)
    g = GpuIndex(nifs, 0)
    if devices:
        g.ref = nifs.flat_new_sharded(0, devices)
    o = oracle_mod.FlatIndex(0)
    shards = nifs.flat_shard_count(g.ref)
    qs = [rng.uniform(-1, 1, d).astype(np.float32) for _ in range(3)]
    seen_chunks = set()
    total = 0
    for step in range(16):
        count = 600 if step < 3 else 2500
        items = [(b"row-%06d" % (total + i), rng.uniform(-1, 1, d).astype(np.float32)) for i in range(count)]
        if step % 4 == 3:   # upserts of early rows and a few deletes between the appends
            items += [(b"row-%06d" % i, rng.uniform(-1, 1, d).astype(np.float32)) for i in range(0, 300, 7)]
        unwrap(nifs.flat_insert_many(g.ref, items))
        o.insert_many(items)
        total += count
        if step % 5 == 4:
            for i in range(5, 200, 13):
                g.delete(b"row-%06d" % i)
                o.delete(b"row-%06d" % i)
        assert len(g) == len(o)
        for s in range(shards):
            n = nifs.flat_shard_lens(g.ref)[s]
            cap, nbytes, chunks = nifs.flat_shard_memory(g.ref, s)
            assert cap >= n and cap * row_bytes <= nbytes
            if chunks:
                assert nbytes == chunks * CHUNK
                assert nbytes < (n + 32) * row_bytes + CHUNK          # never more than the rows plus one chunk
            else:
                assert nbytes < CHUNK + 32 * row_bytes                  # a plain allocation only below one chunk
            seen_chunks.add(chunks)
        _check(g, o, qs + [items[0][1], items[-1][1]])
    assert 0 in seen_chunks and max(seen_chunks) >= (4 if shards == 1 else 2) and len(seen_chunks) >= (4 if shards == 1 else 3)
    # the derived columns and the other searches over a many-chunk slab
    q = qs[0]
    got = unwrap(nifs.flat_search_batch(g.ref, np.stack(qs), 10))
    for i, qq in enumerate(qs):
        assert bits(got[i]) == bits(o.search(qq, 10))
    alive = {h[0] for h in o.search(q, len(o))}
    assert len(alive) == len(o)
    assert bits(unwrap(nifs.flat_quantized_search(g.ref, q, 50, 10)))[0][0] in alive
    # emptying the index and giving it another row width drops the slab
    for id_ in sorted(alive):
        g.delete(id_)
    assert len(g) == 0
    g.insert("fresh", np.ones(200, np.float32))
    assert g.dimension == 200
    owner = int(nifs.flat_route_ids(g.ref, nifs.pack_ids([b"fresh"]))[0])   # (a shard re-dimensions when its next row arrives)
    cap, nbytes, chunks = nifs.flat_shard_memory(g.ref, owner)
    assert chunks == 0 and nbytes < CHUNK
    assert g.search(np.ones(200, np.float32), 1) == [(b"fresh", 0.0)]


def test_forced_plain_allocation_gives_the_same_answers(nifs, oracle_mod, monkeypatch, vt_debug):
    vt_debug.set("slab_chunk_mb", 2)
    d = 96
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (30_000, d)).astype(np.float32)
    ids = [b"v-%d" % i for i in range(len(x))]
    a = GpuIndex(nifs, 2)
    unwrap(nifs.flat_load_matrix(a.ref, ids, x))
    vt_debug.set("slab", 1)
    b = GpuIndex(nifs, 2)
    unwrap(nifs.flat_load_matrix(b.ref, ids, x))
    want_chunks = -(-(30_016 * 128 * 4) // CHUNK)      # rows are padded to 128 floats; capacity in whole 32-row tiles
    assert nifs.flat_shard_memory(a.ref)[2] == want_chunks == 8 and nifs.flat_shard_memory(b.ref)[2] == 0
    for q in (x[0], x[-1], rng.uniform(-1, 1, d).astype(np.float32)):
        assert bits(a.search(q, 25)) == bits(b.search(q, 25))
        assert bits(unwrap(nifs.flat_funnel_search(a.ref, q, [32], 100, 10))) == bits(unwrap(nifs.flat_funnel_search(b.ref, q, [32], 100, 10)))


def test_running_out_of_device_memory_while_growing_is_a_clean_failure(nifs, oracle_mod):
    """An append the card has no room for fails as a whole (the reference's insert_many is
    all-or-nothing, flat.rs:69-85): the rows that were there still answer, the handle is not
    poisoned, the failed allocation does not come back as the next launch's error, and once
    memory is free again the same append goes through."""
    import torch
    d = 768
    rng = np.random.default_rng(8)
    base = rng.uniform(-1, 1, (50_000, d)).astype(np.float32)
    ids = [b"a-%d" % i for i in range(len(base))]
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids, base))
    want = bits(g.search(base[7], 5))
    more = np.tile(rng.uniform(-1, 1, (1000, d)).astype(np.float32), (700, 1))    # 700 000 rows = 2.15 GB: three 1-GiB chunks
    more_ids = [b"b-%d" % i for i in range(len(more))]
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info(0)
    hog = torch.empty(free - (3 << 29), dtype=torch.uint8, device="cuda:0")    # leaves 1.5 GB
    st = nifs.flat_load_matrix(g.ref, more_ids, more)
    assert st[0] == "error" and "memory" in st[1], st
    assert len(g) == len(base)
    assert bits(g.search(base[7], 5)) == want
    assert bits(unwrap(nifs.flat_quantized_search(g.ref, base[7], 20, 5)))[0] == want[0]
    del hog
    torch.cuda.empty_cache()
    unwrap(nifs.flat_load_matrix(g.ref, more_ids, more))
    assert len(g) == len(base) + len(more)
    assert nifs.flat_shard_memory(g.ref)[2] == 3
    assert bits(g.search(base[7], 5)) == want
    assert g.search(more[3], 1)[0][1] == 0.0


def test_large_appends_leave_the_ranking_to_the_next_search(nifs, oracle_mod):
    """A corpus that arrives in several large appends whose ids are not in byte order ("doc-10"
    sorts before "doc-9"): an append that is a small part of what is there does not re-rank all
    ids (a pass over the whole id table per append); the searches that follow -- on the shared
    sentinel rank first, after a re-rank once a staged search asks for strict ranks -- equal
    the oracle's, ties by id included."""
    d = 48
    rng = np.random.default_rng(17)
    o = oracle_mod.FlatIndex(3)
    g = GpuIndex(nifs, 3)
    total = 0
    pool = np.round(rng.uniform(-1, 1, (64, d)) * 2).astype(np.float32) / 2     # coarse rows: many exact ties
    for step, count in enumerate((120_000, 20_000, 25_000, 17_000)):
        x = rng.uniform(-1, 1, (count, d)).astype(np.float32)
        x[::50] = pool[rng.integers(0, 64, size=len(x[::50]))]
        ids = [b"doc-%d" % (total + i + 1) for i in range(count)]
        unwrap(nifs.flat_load_matrix(g.ref, ids, x))
        o.insert_matrix(ids, x)
        total += count
        for q in (pool[3], pool[40], x[5], rng.uniform(-1, 1, d).astype(np.float32)):
            for k in (1, 10, 100):
                assert bits(g.search(q, k)) == bits(o.search(q, k)), (step, k)
    q = pool[7]
    sign = oracle_mod.compress_sign_bits
    assert len(g) == len(o) == total
    got = unwrap(nifs.flat_quantized_search(g.ref, q, 64, 10))     # strict ranks: the deferred re-rank happens here
    rows = {h[0] for h in got}
    assert len(rows) == 10
    assert bits(g.search(q, 300)) == bits(o.search(q, 300))
    del sign


@pytest.mark.parametrize("d", [64, 40])
def test_rows_from_another_device_reach_a_mapped_slab_through_a_staging_block(nifs, oracle_mod, monkeypatch, d, request, vt_debug):
    """Only the owning device is given access to a mapped slab's chunks, so rows resident on
    another GPU of the node are first copied into an ordinary buffer on this one and placed from
    there (`VT_TEST_FOREIGN_ROWS` makes the one GPU of this box count as "another"): appends,
    scattered upserts, an id twice in a batch (the last one wins, flat.rs:270-281); d = 40 also
    pads the rows on the way."""
    if support.rerun_with_hooks_library(request):   # (VT_TEST_FOREIGN_ROWS only exists in libvettore_hip_hooks.so)
        return
    import torch
    vt_debug.set("slab_chunk_mb", 2)
    vt_debug.set("test_foreign_rows", 1)
    rng = np.random.default_rng(3)
    n = 30_000
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    ids = [b"r-%05d" % i for i in range(n)]
    g = GpuIndex(nifs, 0)
    o = oracle_mod.FlatIndex(0)
    xd = torch.from_numpy(x).to("cuda:0")
    assert nifs.flat_load_device_matrix(g.ref, nifs.pack_ids(ids), xd.data_ptr(), n, d) == ("ok", ())
    o.insert_matrix(ids, x)
    assert nifs.flat_shard_memory(g.ref)[2] >= 3
    pick = rng.permutation(n)[:4000]
    up_ids = [ids[i] for i in pick] + [ids[int(pick[0])]] + [b"new-%d" % i for i in range(300)]
    up = rng.uniform(-1, 1, (len(up_ids), d)).astype(np.float32)
    ud = torch.from_numpy(up).to("cuda:0")
    assert nifs.flat_load_device_matrix(g.ref, nifs.pack_ids(up_ids), ud.data_ptr(), len(up_ids), d) == ("ok", ())
    o.insert_many(list(zip(up_ids, up)))
    assert len(g) == len(o) == n + 300
    for q in (up[0], up[len(pick)], up[-1], x[1], rng.uniform(-1, 1, d).astype(np.float32)):
        assert bits(g.search(q, 20)) == bits(o.search(q, 20))
    assert g.search(up[len(pick)], 1)[0] == (ids[int(pick[0])], 0.0)
