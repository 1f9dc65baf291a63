"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through
the C ABI of libvettore_hip.so, against (a) the reference's golden fixtures and
(b) the CPU oracle on the same seeded inputs.

Bar: ids/ranking identical AND raw scores bit-identical (the kernels reproduce
the reference's f32 operation order exactly, for each selectable lane order of
wide::f32x8::reduce_add), so no tolerance is needed; the 1e-5 relative bound of
BASELINE.json is asserted as well where an independent f64 value exists.
"""
import math

import numpy as np
import pytest

import support
from support import b, close, load, run_steps, same_f32

pytestmark = pytest.mark.gpu

ORDERS = [0, 1, 2, 3]
TOL = 1e-5  # BASELINE.json: f32 scores within 1e-5 (relative, distances.rs:485-491 convention)


class GpuError(Exception):
    pass


@pytest.fixture(scope="module")
def nifs():
    from vettore_amd import nifs as n
    import vettore_amd._lib as L
    assert L.load().vt_device_count() >= 1, "no HIP device: GPU tests need the real hardware"
    return n


def unwrap(res):
    if res == "ok":
        return None
    if res[0] == "ok":
        return res[1]
    raise GpuError(res[1])


class GpuIndex:
    """support.run_steps adapter over the Python mirror of Vettore.Nifs."""

    def __init__(self, nifs, metric_code, order=3):
        self.n = nifs
        self.ref = nifs._flat_new(metric_code)
        nifs.flat_set_reduce_order(self.ref, order)

    def insert(self, id_, vector):
        return unwrap(self.n.flat_insert(self.ref, id_, vector))

    def insert_many(self, items):
        return unwrap(self.n.flat_insert_many(self.ref, items))

    def delete(self, id_):
        return unwrap(self.n.flat_delete(self.ref, id_))

    def search(self, query, limit):
        return unwrap(self.n.flat_search(self.ref, query, limit))

    def __len__(self):
        return len(self.ref)

    @property
    def dimension(self):
        return self.ref.dimension


def bits(hits):
    return [(h[0], np.float32(h[1]).tobytes()) for h in hits]


# ------------------------------------------------- reference golden fixtures
def test_flat_rs_scripts(nifs, oracle_mod):
    for case in load("flat_rs.json"):
        if case.get("differential"):
            continue
        ix = GpuIndex(nifs, oracle_mod.METRIC_CODE[case["metric"]])
        run_steps(ix, case["steps"], GpuError)


@pytest.mark.parametrize("order", ORDERS)
def test_flat_rs_all_metrics_match_oracle_bitwise(nifs, oracle_mod, order):
    case = next(c for c in load("flat_rs.json") if c.get("differential"))
    oracle_mod.set_reduce_order(order)
    try:
        for name in case["metrics"]:
            m = oracle_mod.METRIC_CODE[name]
            g = GpuIndex(nifs, m, order)
            o = oracle_mod.FlatIndex(m)
            items = [(r[0], r[1]) for r in case["rows"]]
            g.insert_many(items)
            o.insert_many(items)
            for limit in case["limits"]:
                assert bits(g.search(case["query"], limit)) == bits(o.search(case["query"], limit)), (name, limit)
    finally:
        oracle_mod.set_reduce_order(oracle_mod.DEFAULT_ORDER)


def test_distances_tail_lengths_bitwise(nifs, oracle_mod):
    """distances.rs:570-609: chunked kernel + scalar tail for len 1..40 (len 0 is
    rejected by the index: "vector must not be empty")."""
    c = load("distances_rs.json")["simd_and_tail_kernels_match_scalar_oracles"]
    for order in ORDERS:
        oracle_mod.set_reduce_order(order)
        for v in c["vectors"]:
            if v["len"] == 0:
                continue
            for m in (0, 1, 2, 3, 4, 5, 6, 7, 8):
                g = GpuIndex(nifs, m, order)
                g.insert("x", v["right"])
                got = g.search(v["left"], 1)
                want = oracle_mod.compute(m, v["left"], v["right"])
                assert same_f32(got[0][1], want), (order, m, v["len"])
                exact = None
                l64 = np.asarray(v["left"], dtype=np.float64)
                r64 = np.asarray(v["right"], dtype=np.float64)
                if m == 3:
                    exact = float(np.sum(l64 * r64))
                elif m == 1:
                    exact = float(np.sum((l64 - r64) ** 2))
                elif m == 5:
                    exact = float(np.sum(np.abs(l64 - r64)))
                if exact is not None:
                    assert close(got[0][1], exact, c["tolerance"])
    oracle_mod.set_reduce_order(oracle_mod.DEFAULT_ORDER)


def test_distances_overflow_recovery(nifs, oracle_mod):
    c = load("distances_rs.json")["recovers_representable_results_after_f32_intermediate_overflow"]
    code = oracle_mod.METRIC_CODE
    for name, l, r, want, tol in c["close"]:
        g = GpuIndex(nifs, code[name])
        g.insert("x", r)
        got = g.search(l, 1)[0][1]
        assert close(got, want, tol) and same_f32(got, oracle_mod.compute(code[name], l, r))
    for name, l, r, want, sign in c["exact"]:
        g = GpuIndex(nifs, code[name])
        g.insert("x", r)
        got = g.search(l, 1)[0][1]
        assert same_f32(got, want), (name, got)
    for name, l, r in c["errors"]:
        g = GpuIndex(nifs, code[name])
        g.insert("x", r)
        with pytest.raises(GpuError, match="metric overflow"):
            g.search(l, 1)


def test_normalize_and_sign_bits(nifs, oracle_mod):
    d = load("distances_rs.json")
    c = d["validates_dimensions_normalization_and_finite_values"]
    for v, want in c["normalize_l2"]:
        assert list(unwrap(nifs.normalize_l2(v))) == [np.float32(x) for x in want]
    for v, want, tol in c["normalize_l2_close"]:
        got = unwrap(nifs.normalize_l2(v))
        assert all(abs(float(g) - w) < tol for g, w in zip(got, want))
    for bad in d["cosine_and_normalization_obey_numerical_invariants"]["non_finite"]:
        assert nifs.normalize_l2([bad]) == ("error", "vector contains a non-finite value")
    c = d["packs_bits_and_masks_unused_coordinates"]
    for v, words in c["compress"]:
        assert nifs.compress_sign_bits(v) == words
    rng = np.random.default_rng(11)
    for n in (1, 7, 63, 64, 65, 200, 768, 1000):
        v = rng.uniform(-1, 1, n).astype(np.float32)
        v[rng.integers(0, n)] = 0.0
        if n > 2:
            v[1] = -0.0
        assert np.array_equal(unwrap(nifs.normalize_l2(v)).view(np.uint32), oracle_mod.normalize_l2(v).view(np.uint32)), n
        assert nifs.compress_sign_bits(v) == [int(w) for w in oracle_mod.compress_sign_bits(v)], n


# ---------------------------------------------------------------- search.rs
def _expect_call(res, call):
    if "expect_error" in call:
        assert res == ("error", call["expect_error"]), res
        return
    hits = unwrap(res)
    if "expect" in call:
        assert hits == [(b(e[0]), e[1]) for e in call["expect"]]
    if "expect_first_id" in call:
        assert hits[0][0] == b(call["expect_first_id"])


def test_search_rs_vector_top_k(nifs, oracle_mod):
    d = load("search_rs.json")
    code = oracle_mod.METRIC_CODE
    c = d["vector_top_k_handles_prefixes_similarity_and_ties"]
    for call in c["calls"]:
        _expect_call(nifs.vector_top_k([(v[0], v[1]) for v in c["vectors"]], call["query"], code[call["metric"]],
                                       call["dimensions"], call["limit"]), call)
    for key in ("vector_top_k_rejects_bad_dimensions_and_values",
                "vector_top_k_validates_queries_and_only_reads_the_requested_prefix"):
        for call in d[key]["calls"]:
            _expect_call(nifs.vector_top_k([(v[0], v[1]) for v in call["vectors"]], call["query"],
                                           code[call["metric"]], call["dimensions"], call["limit"]), call)
    c = d["stable_ties_do_not_depend_on_candidate_order"]
    for vectors in (c["forward"], list(reversed(c["forward"]))):
        _expect_call(nifs.vector_top_k([(v[0], v[1]) for v in vectors], c["query"], code[c["metric"]],
                                       c["dimensions"], c["limit"]), c)


def test_search_rs_vector_top_k_full_grid_bitwise(nifs, oracle_mod):
    c = load("search_rs.json")["vector_top_k_matches_full_sort_for_every_metric_and_limit"]
    rows = [(r[0], r[1]) for r in c["rows"]]
    for name in c["metrics"]:
        m = oracle_mod.METRIC_CODE[name]
        for dims in c["dimensions"]:
            for limit in c["limits"]:
                got = unwrap(nifs.vector_top_k(rows, c["query"], m, dims, limit))
                want = oracle_mod.vector_top_k(rows, c["query"], m, dims, limit)
                assert bits(got) == bits(want), (name, dims, limit)


def test_distances_rs_packed_word_boundaries_on_device(nifs):
    """distances.rs:675-707 through binary_top_k on the GPU: dims {1,63,64,65,127,128,129}, the
    tail word of one row filled with dirty padding bits -- the distance counts only the
    flipped coordinates (word_mask, distances.rs:459-481)."""
    c = load("distances_rs.json")["packed_distances_cover_word_boundaries_and_ignore_padding"]
    full = (1 << 64) - 1
    for dims in c["dimensions"]:
        words = (dims + 63) // 64
        left = [full] * words
        right = list(left)
        flipped = [0] + ([dims - 1] if dims > 1 else [])
        for coord in flipped:
            right[coord // 64] ^= 1 << (coord % 64)
        if dims % 64:
            used = (1 << (dims % 64)) - 1
            right[words - 1] ^= (~used) & full           # dirty padding
        got = unwrap(nifs.binary_top_k([("same", left), ("flipped", right)], left, dims, 2))
        assert got == [(b"same", 0.0), (b"flipped", float(len(flipped)))], dims
        # dirty padding in the QUERY must be ignored as well
        got = unwrap(nifs.binary_top_k([("same", left), ("flipped", right)], right, dims, 2))
        assert got == [(b"flipped", 0.0), (b"same", float(len(flipped)))], dims
    res = nifs.binary_top_k([("a", [])], [], 1, 1)        # distances.rs:704-706: no words for 1 dimension
    assert res == ("error", "dimension mismatch")


def test_distances_rs_cosine_edge_cases_on_device(nifs, oracle_mod):
    """distances.rs:637-673 (`cosine`: f64 dot / f64 norms, clamp, zero norm => 0.0) through
    vector_top_k(metric = cosine) on the GPU, search.rs:56-60."""
    c = load("distances_rs.json")["cosine_and_normalization_obey_numerical_invariants"]
    for l, r, want, tol in c["cosine_close"]:
        got = unwrap(nifs.vector_top_k([("x", r)], l, 2, len(l), 1))
        assert got[0][0] == b"x" and close(got[0][1], want, tol)
        assert same_f32(got[0][1], oracle_mod.cosine(l, r))
    # zero norm on either side => 0.0 (distances.rs:168-170)
    for l, r in (([0.0, 0.0], [1.0, 2.0]), ([1.0, 2.0], [0.0, 0.0]), ([0.0], [0.0])):
        got = unwrap(nifs.vector_top_k([("z", r)], l, 2, len(l), 1))
        assert same_f32(got[0][1], 0.0) and same_f32(got[0][1], oracle_mod.cosine(l, r))
    # dimension mismatch: a row shorter than the prefix (search.rs:49-55)
    l, r, msg = c["cosine_errors"][0]
    assert nifs.vector_top_k([("short", l)], r, 2, len(r), 1) == ("error", msg)
    for bad in c["non_finite"]:
        assert nifs.vector_top_k([("x", [1.0])], [bad], 2, 1, 1) == ("error", "vector contains a non-finite value")
        assert nifs.vector_top_k([("x", [bad])], [1.0], 2, 1, 1) == ("error", "vector contains a non-finite value")
    # the clamp: parallel vectors whose f64 quotient lands a hair above 1.0
    v = [0.1, 0.2, 0.3, 0.7]
    got = unwrap(nifs.vector_top_k([("p", [3.0 * t for t in v])], v, 2, 4, 1))
    assert got[0][1] <= 1.0 and same_f32(got[0][1], oracle_mod.cosine(v, [3.0 * t for t in v]))


def test_search_rs_binary_top_k(nifs, oracle_mod):
    d = load("search_rs.json")
    c = d["binary_top_k_masks_padding_and_orders_ids"]
    q = nifs.compress_sign_bits(c["query_vector"])
    vecs = [(v[0], nifs.compress_sign_bits(v[1])) for v in c["vectors"]]
    assert unwrap(nifs.binary_top_k(vecs, q, c["dimensions"], c["limit"])) == [(b(e[0]), e[1]) for e in c["expect"]]
    for call in d["binary_top_k_validates_empty_batches_limits_and_word_boundaries"]["calls"]:
        _expect_call(nifs.binary_top_k([(v[0], v[1]) for v in call["vectors"]], call["query"], call["dimensions"],
                                       call["limit"]), call)


# ----------------------------------------- Elixir tests through the NIF surface
def test_elixir_all_metrics_through_collection(nifs):
    from vettore_amd.collection import Collection
    c = load("elixir_nif.json")["all_supported_metrics_return_stable_top_k_results"]
    for name in c["metrics"]:
        ok, col = Collection.new(dimensions=2, metric=name, index="flat")
        assert ok == "ok"
        assert col.put_many([{"id": r[0], "vector": r[1]} for r in c["rows"]]) == "ok"
        ok, results = col.search(c["query"], {"limit": c["limit"]})
        assert ok == "ok" and [r.id for r in results] == [b(x) for x in c["expect_ids"]], name
        assert col.close() == "ok"


def test_elixir_phantom_id_and_ok_unit(nifs):
    from vettore_amd.collection import Collection
    from vettore_amd.index_flat import FlatGpu
    from vettore_amd.collection import Embedding
    c = load("elixir_nif.json")["phantom_native_id_and_ok_unit"]
    ok, col = Collection.new(dimensions=1, metric="l2")
    assert FlatGpu.put(col, Embedding(id="bad", vector=[])) == ("error", c["put_empty_error"])
    assert nifs.flat_insert(col.index_state, c["flat_insert"][0], c["flat_insert"][1]) == ("ok", ())
    assert col.search([0.0], {"limit": 1}) == ("ok", [])
    assert FlatGpu.new("l2", {"unknown": True}) == ("error", "invalid_flat_options")
    assert FlatGpu.new("unknown", []) == ("error", ("unsupported_flat_metric", "unknown"))
    assert FlatGpu.search(col, [0.0], {"unknown": True}) == ("error", "invalid_search_options")


def test_elixir_batched_helpers(nifs):
    c = load("elixir_nif.json")["batched_native_helpers"]
    vecs = [(v[0], v[1]) for v in c["vectors"]]
    for mc in c["metric_codes"]:
        hits = unwrap(nifs.vector_top_k(vecs, c["query"], mc, c["dimensions"], c["limit"]))
        assert [h[0] for h in hits] == [b(x) for x in c["expect_ids"]], mc
    assert nifs.vector_top_k(vecs, c["query"], c["unknown_metric"][0], 2, 2) == ("error", c["unknown_metric"][1])
    assert nifs.vector_top_k(vecs, c["query"], 0, 0, 2) == ("error", c["bad_prefix"][2])
    bn = c["binary"]
    assert unwrap(nifs.binary_top_k([(v[0], v[1]) for v in bn["vectors"]], bn["query"], bn["dimensions"],
                                    bn["limit"])) == [(b(e[0]), e[1]) for e in bn["expect"]]


def test_elixir_cosine_collection_and_result_values(nifs):
    from vettore_amd.collection import Collection
    from vettore_amd.index_flat import result_values
    d = load("elixir_nif.json")
    c = d["cosine_collection_result_semantics"]
    ok, col = Collection.new(dimensions=2, metric="cosine", normalize="l2", score="raw")
    assert col.put_many([{"id": r[0], "vector": r[1]} for r in c["rows"]]) == "ok"
    assert col.put({"id": "right", "vector": [0.5, 0.5]}) == ("error", "duplicate_id")
    ok, results = col.search(c["query"], {"limit": c["limit"]})
    first = results[0]
    assert (first.id, first.score, first.distance, first.metric) == (b"right", 1.0, 0.0, "cosine")
    for metric, raw, mode, want in d["result_values"]["table"]:
        assert list(result_values(metric, raw, mode)) == want
    a = d["adapter_validation"]
    ok, col = Collection.new(dimensions=a["dimensions"], metric=a["metric"])
    for lim in a["invalid_limits"]:
        assert col.search([0.0, 0.0], {"limit": lim}) == ("error", "invalid_limit")
    assert col.search(a["dimension_mismatch_query"], {"limit": 1}) == ("error", "dimension_mismatch")


def test_elixir_quantized_search(nifs):
    from vettore_amd.collection import Collection
    d = load("elixir_nif.json")
    c = d["binary_quantized_search"]
    ok, col = Collection.new(dimensions=2, metric=c["metric"], index="flat")
    assert col.put_many([{"id": r[0], "vector": r[1]} for r in c["rows"]]) == "ok"
    assert col.get(c["binary_vector_of"][0])[1].binary_vector == c["binary_vector_of"][1]
    ok, results = col.quantized_search(c["query"], {"candidates": c["candidates"], "limit": c["limit"]})
    assert [(r.id, r.distance) for r in results] == [(b(e["id"]), e["distance"]) for e in c["expect"]]
    c = d["full_candidate_adaptive_modes_agree_with_exact_flat_search"]
    ok, col = Collection.new(dimensions=4, metric=c["metric"], index="flat")
    assert col.put_many([{"id": r[0], "vector": r[1]} for r in c["rows"]]) == "ok"
    ok, exact = col.search(c["query"], {"limit": c["limit"]})
    ok, quant = col.quantized_search(c["query"], {"candidates": c["candidates"], "limit": c["limit"]})
    assert [r.id for r in quant] == [r.id for r in exact]


# ------------------------------------------- seeded random parity vs the oracle
def make_corpus(n, d, seed, normalize, oracle_mod, dup_frac=0.01, tie_block=0):
    """iid uniform(-1,1), 1% verbatim duplicate rows, ids
    "doc-<i>" (bytewise order != numeric order)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.0, 1.0, size=(n, d)).astype(np.float32)
    ndup = int(n * dup_frac)
    if ndup:
        src = rng.integers(0, n, ndup)
        dst = rng.integers(0, n, ndup)
        x[dst] = x[src]
    if tie_block:
        x[n // 2:n // 2 + tie_block] = x[n // 2]
    if normalize:
        x = np.stack([oracle_mod.normalize_l2(r) for r in x])
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    return x, ids


@pytest.mark.parametrize("d", [8, 24, 100, 7, 200, 384])
@pytest.mark.parametrize("order", ORDERS)
def test_random_parity_all_metrics(nifs, oracle_mod, d, order):
    n = 5000
    oracle_mod.set_reduce_order(order)
    try:
        x, ids = make_corpus(n, d, 20260721 + d, False, oracle_mod, tie_block=40)
        x[:, d // 2] = np.where(np.arange(n) % 3 == 0, 0.0, x[:, d // 2])  # exercise hamming/jaccard zeros
        packed = oracle_mod.pack_ids(ids)
        rng = np.random.default_rng(20260722)
        queries = rng.uniform(-1, 1, size=(3, d)).astype(np.float32)
        queries[0] = x[n // 2]  # hits the identical-row block exactly
        for m in range(9):
            g = GpuIndex(nifs, m, order)
            unwrap(nifs.flat_load_matrix(g.ref, ids, x))
            for q in queries:
                for k in (1, 10, 64, 100, 300):
                    got = g.search(q, k)
                    want = oracle_mod.matrix_search(m, x, packed, q, k)
                    assert bits(got) == bits(want), (d, order, m, k)
    finally:
        oracle_mod.set_reduce_order(oracle_mod.DEFAULT_ORDER)


@pytest.mark.parametrize("d", [1024, 1536, 1540, 3072, 4100, 8192])
def test_wide_rows_use_column_panels(nifs, oracle_mod, d):
    """Rows wider than one LDS panel (96 chunks) are walked in column panels with
    the running sum carried across them; the tail chunk lands in the last one."""
    n = 1500
    x, ids = make_corpus(n, d, 99 + d, False, oracle_mod, tie_block=20)
    packed = oracle_mod.pack_ids(ids)
    rng = np.random.default_rng(d)
    q = rng.uniform(-1, 1, d).astype(np.float32)
    for order in ORDERS:
        oracle_mod.set_reduce_order(order)
        try:
            for m in ((0, 2, 3, 5, 6, 7) if d < 8192 else (2, 0)):
                g = GpuIndex(nifs, m, order)
                unwrap(nifs.flat_load_matrix(g.ref, ids, x))
                for k in (10, 100):
                    assert bits(g.search(q, k)) == bits(oracle_mod.matrix_search(m, x, packed, q, k)), (d, order, m, k)
        finally:
            oracle_mod.set_reduce_order(oracle_mod.DEFAULT_ORDER)


def test_cosine_config1_parity(nifs, oracle_mod):
    """BASELINE.json configs[0]: flat cosine, d=384, N=10k, limit=10 (the
    reference's own CPU-runnable case) + the tie set of SURVEY.md 8d."""
    n, d = 10000, 384
    x, ids = make_corpus(n, d, 20260721, True, oracle_mod, tie_block=64)
    packed = oracle_mod.pack_ids(ids)
    g = GpuIndex(nifs, 2)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(20260722)
    for i in range(20):
        q = oracle_mod.normalize_l2(rng.uniform(-1, 1, d).astype(np.float32))
        if i == 0:
            q = x[n // 2]
        got = g.search(q, 10 if i else 80)
        want = oracle_mod.matrix_search(2, x, packed, q, 10 if i else 80)
        assert bits(got) == bits(want), i
        exact = x.astype(np.float64) @ q.astype(np.float64)
        by_id = {ids[j]: exact[j] for j in range(n)}
        assert all(close(h[1], by_id[h[0]], TOL) for h in got)


def test_mutations_follow_the_oracle(nifs, oracle_mod):
    """Random insert / upsert / delete / search interleaving (flat.rs:59-93):
    swap-delete, slab growth and id-rank maintenance must stay invisible."""
    rng = np.random.default_rng(5)
    d = 16
    for m in (0, 2, 3):
        g = GpuIndex(nifs, m)
        o = oracle_mod.FlatIndex(m)
        live = []
        for step in range(400):
            op = rng.integers(0, 10)
            if op < 5 or not live:
                cnt = int(rng.integers(1, 40))
                items = [("id-%d" % rng.integers(0, 600), rng.uniform(-1, 1, d).astype(np.float32)) for _ in range(cnt)]
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
            q = rng.uniform(-1, 1, d).astype(np.float32)
            k = int(rng.integers(1, 30))
            if len(o) == 0:
                continue
            got, want = g.search(q, k), o.search(q, k)
            if bits(got) != bits(want):  # say what kind of difference it is before failing
                full_g, full_o = g.search(q, len(o)), o.search(q, len(o))
                gd, od = dict(full_g), dict(full_o)
                first = next(i for i, (a_, b_) in enumerate(zip(got, want)) if bits([a_]) != bits([b_]))
                detail = {"k": k, "n": len(o), "first_diff": first, "gpu": got[first], "oracle": want[first],
                          "gpu_raw_of_wanted_id": gd.get(want[first][0]), "oracle_raw_of_gpu_id": od.get(got[first][0]),
                          "full_lists_equal": bits(full_g) == bits(full_o), "second_try_equal": bits(g.search(q, k)) == bits(want)}
                raise AssertionError((m, step, detail))


def test_sorted_and_unsorted_id_arrival(nifs, oracle_mod):
    """id-rank fast path (ascending appends) and the re-sort path agree."""
    rng = np.random.default_rng(9)
    d, n = 8, 3000
    x = np.round(rng.uniform(-1, 1, size=(n, d)) * 4).astype(np.float32) / 4  # many exact ties
    sorted_ids = sorted(b"k%05d" % i for i in range(n))
    shuffled = list(sorted_ids)
    rng.shuffle(shuffled)
    q = x[0]
    outs = []
    for ids in (sorted_ids, shuffled):
        g = GpuIndex(nifs, 0)
        for s in range(0, n, 500):
            g.insert_many([(ids[i], x[i]) for i in range(s, s + 500)])
        outs.append((bits(g.search(q, 200)), bits(oracle_mod.matrix_search(0, x, oracle_mod.pack_ids(ids), q, 200))))
    assert outs[0][0] == outs[0][1] and outs[1][0] == outs[1][1]


@pytest.mark.parametrize("metric", [0, 2, 3])
def test_quantized_search_matches_oracle_composition(nifs, oracle_mod, metric):
    """collection.ex:276-295 = binary_top_k over compress_sign_bits(rows) then
    vector_top_k (f64 cosine for metric 2) -- composed from oracle pieces."""
    n, d = 4000, 200
    x, ids = make_corpus(n, d, 77 + metric, metric == 2, oracle_mod)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(3)
    obits = [(ids[i], oracle_mod.compress_sign_bits(x[i])) for i in range(n)]
    by_id = {ids[i]: x[i] for i in range(n)}
    for cand, limit in ((100, 10), (300, 20), (n, 10)):
        q = rng.uniform(-1, 1, d).astype(np.float32)
        if metric == 2:
            q = oracle_mod.normalize_l2(q)
        c = oracle_mod.binary_top_k(obits, oracle_mod.compress_sign_bits(q), d, cand)
        want = oracle_mod.vector_top_k([(cid, by_id[cid]) for cid, _ in c], q, metric, d, limit)
        got = unwrap(nifs.flat_quantized_search(g.ref, q, cand, limit))
        assert bits(got) == bits(want), (metric, cand, limit)


@pytest.mark.parametrize("shape", ["random", "two_patterns", "all_equal"])
def test_quantized_search_histogram_pass(nifs, oracle_mod, shape):
    """Above 16 384 rows the candidate pass is the histogram/threshold stream (K4h).  Random
    rows; rows drawn from two sign patterns (thousands of ties at the k-th distance, broken by
    id bytes); and 150 000 identical rows (more ties than the device list holds: the call must
    fall back to the list-carrying scan and still agree with the oracle)."""
    d = 64
    rng = np.random.default_rng(515)
    if shape == "random":
        n = 40_000
        x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    elif shape == "two_patterns":
        n = 30_000
        base = rng.uniform(-1, 1, (2, d)).astype(np.float32)
        x = base[rng.integers(0, 2, n)] * rng.uniform(0.5, 1.5, (n, 1)).astype(np.float32)
    else:
        n = 150_000
        x = np.tile(rng.uniform(-1, 1, (1, d)).astype(np.float32), (n, 1))
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    packed = oracle_mod.pack_ids(ids)
    xbits = np.stack([oracle_mod.compress_sign_bits(r) for r in x[:2]]) if shape == "all_equal" else None
    # 257..4096 candidates: the candidate set stays on the device as an unsorted list;
    # above that the host-staged multi-pass path
    cases = ((100, 10), (256, 30), (7, 7)) + (((300, 20), (1000, 100), (4096, 10), (5000, 10)) if shape != "all_equal" else ((1000, 10),))
    for cand, limit in cases:
        q = rng.uniform(-1, 1, d).astype(np.float32)
        qb = oracle_mod.compress_sign_bits(q)
        # oracle composition without materialising n Python tuples: distances via numpy popcount
        sign = (x >= 0)
        ham = (sign != (q >= 0)[None, :]).sum(axis=1)
        order = sorted(range(n), key=lambda i: (int(ham[i]), ids[i]))[:cand] if n <= 40_000 else \
            sorted(np.argsort(ham, kind="stable")[:max(cand * 50, 10_000)].tolist() if shape != "all_equal" else range(n),
                   key=lambda i: (int(ham[i]), ids[i]))[:cand]
        assert oracle_mod.packed_hamming(oracle_mod.compress_sign_bits(x[order[0]]), qb, d) == ham[order[0]]
        want = oracle_mod.vector_top_k([(ids[i], x[i]) for i in order], q, 0, d, limit)
        got = unwrap(nifs.flat_quantized_search(g.ref, q, cand, limit))
        assert bits(got) == bits(want), (shape, cand, limit)


def test_funnel_overflow_in_any_stage_is_reported(nifs, oracle_mod):
    """search.rs:38-73 returns Err("metric overflow") whichever stage meets the row; the
    chained device path carries the flag from an intermediate stage to the final select."""
    n, d = 300, 16
    rng = np.random.default_rng(77)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    ids = [b"r%03d" % i for i in range(n)]
    q = np.full(d, 2.0, np.float32)
    early, late = x.copy(), x.copy()
    early[7, :] = 3e38                 # overflows already on the 8-wide prefix
    late[9, :8] = 100.0                # best prefix score, so it survives stage 1 ...
    late[9, 8:] = 3e38                 # ... and overflows in the full-width rerank
    for m in (early, late):
        g = GpuIndex(nifs, 3)
        unwrap(nifs.flat_load_matrix(g.ref, ids, m))
        rows = [(ids[i], m[i]) for i in range(n)]
        for cand in (50, 300):         # device chain / host-staged path
            with pytest.raises(oracle_mod.OracleError, match="metric overflow"):
                kept = oracle_mod.vector_top_k(rows, q, 3, 8, cand)
                by_id = dict(rows)
                oracle_mod.vector_top_k([(i, by_id[i]) for i, _ in kept], q, 3, d, 5)
            assert nifs.flat_funnel_search(g.ref, q, [8], cand, 5) == ("error", "metric overflow")
        # the flag does not leak into the next call
        ok = nifs.flat_funnel_search(g.ref, np.zeros(d, np.float32), [8], 50, 5)
        assert ok[0] == "ok"


@pytest.mark.parametrize("metric", [2, 0, 3])
def test_funnel_search_matches_oracle_composition(nifs, oracle_mod, metric):
    """collection.ex:245-260, :674-691: per stage vector_top_k on a prefix (f64
    cosine for metric 2), first stage over the whole corpus, then exact rerank --
    composed from oracle pieces, compared bit for bit."""
    n, d = 6000, 200
    x, ids = make_corpus(n, d, 310 + metric, metric == 2, oracle_mod, tie_block=30)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(4)
    rows = [(ids[i], x[i]) for i in range(n)]
    for stages, cand, limit in (([64], 100, 10), ([13, 100], 50, 5), ([128, 200], 300, 20), ([200], 10, 10)):
        q = rng.uniform(-1, 1, d).astype(np.float32)
        if metric == 2:
            q = oracle_mod.normalize_l2(q)
        cur = rows
        for st in stages:
            kept = oracle_mod.vector_top_k(cur, q, metric, st, cand)
            by_id = dict(cur)
            cur = [(i, by_id[i]) for i, _ in kept]
        want = oracle_mod.vector_top_k(cur, q, metric, d, limit)
        got = unwrap(nifs.flat_funnel_search(g.ref, q, stages, cand, limit))
        assert bits(got) == bits(want), (metric, stages, cand, limit)
    assert nifs.flat_funnel_search(g.ref, q, [0], 10, 5) == ("error", "invalid prefix dimensions")
    assert nifs.flat_funnel_search(g.ref, q, [d + 1], 10, 5) == ("error", "invalid prefix dimensions")
    assert nifs.flat_funnel_search(g.ref, q, [], 10, 5) == ("error", "invalid prefix dimensions")


def test_elixir_funnel_equals_flat_with_full_candidates(nifs):
    """test/vector_adversarial_test.exs:376-421: funnel_search(stages: [2, 4], candidates: 64) ids == flat ids."""
    from vettore_amd.collection import Collection
    c = load("elixir_nif.json")["full_candidate_adaptive_modes_agree_with_exact_flat_search"]
    ok, col = Collection.new(dimensions=4, metric=c["metric"], index="flat")
    assert col.put_many([{"id": r[0], "vector": r[1]} for r in c["rows"]]) == "ok"
    ok, exact = col.search(c["query"], {"limit": c["limit"]})
    ok, funnel = col.funnel_search(c["query"], {"stages": [2, 4], "candidates": c["candidates"], "limit": c["limit"]})
    assert [r.id for r in funnel] == [r.id for r in exact]
    assert col.funnel_search(c["query"], {"stages": [5]}) == ("error", "invalid_stages")
    ok, hybrid = col.hybrid_search(c["query"], {"generators": [("funnel", {"stages": [2, 4], "candidates": 64}),
                                                                ("quantized", {"candidates": 64}),
                                                                ("search", {"candidates": 64})], "limit": c["limit"]})
    assert [r.id for r in hybrid] == [r.id for r in exact]


@pytest.mark.parametrize("metric", [2, 0, 5, 7])
def test_short_chains_on_a_corpus_of_few_blocks(nifs, oracle_mod, metric):
    """flat_search on a corpus of few blocks, quantized_search, funnel_search (incl. the sign / non-zero bits that ride
    behind the query's floats in one copy): the oracle's hits.  (r05 also read the query in place from the pinned block
    here, `direct_query`; that form measured no gain and has left the library.)"""
    n, d = 3000, 200
    x, ids = make_corpus(n, d, 900 + metric, metric == 2, oracle_mod, tie_block=20)
    if metric == 7:
        x[np.random.default_rng(5).uniform(size=x.shape) < 0.5] = 0.0
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    packed = oracle_mod.pack_ids(ids)
    rng = np.random.default_rng(77)
    rows = list(zip(ids, x))
    for step in range(6):
        q = x[int(rng.integers(0, n))].copy() if step % 3 == 0 else rng.uniform(-1, 1, d).astype(np.float32)
        if metric == 2:
            q = oracle_mod.normalize_l2(q)
        got = {0: (bits(unwrap(nifs.flat_search(g.ref, q, 25))),
                   bits(unwrap(nifs.flat_quantized_search(g.ref, q, 100, 10))),
                   bits(unwrap(nifs.flat_funnel_search(g.ref, q, [64, 128], 100, 10))))}
        assert got[0][0] == bits(oracle_mod.matrix_search(metric, x, packed, q, 25)), (metric, step)
        cands = oracle_mod.binary_top_k([(i, oracle_mod.compress_sign_bits(v)) for i, v in rows], oracle_mod.compress_sign_bits(q), d, 100)
        by_id = dict(rows)
        assert got[0][1] == bits(oracle_mod.vector_top_k([(i, by_id[i]) for i, _ in cands], q, metric, d, 10)), (metric, step)
        cur = rows
        for st in (64, 128):
            cur = [(i, by_id[i]) for i, _ in oracle_mod.vector_top_k(cur, q, metric, st, 100)]
        assert got[0][2] == bits(oracle_mod.vector_top_k(cur, q, metric, d, 10)), (metric, step)


def test_adapter_staged_searches_their_batches_and_their_errors(nifs):
    """Vettore.Index.FlatGpu's quantized / funnel / hybrid wrappers and their batch forms (integration/lib/vettore/index/
    flat_gpu.ex, mirrored in vettore_amd/index_flat.py), reached through the collection's dispatch (INTEGRATION.md
    section 3): the scenario of test/vector_adversarial_test.exs:376-421, every batched list equal to the single call,
    error atoms and their order as run_hybrid_generator's (collection.ex:536-556, :1136-1142)."""
    from vettore_amd.collection import Collection
    from vettore_amd.index_flat import FlatGpu
    c = load("elixir_nif.json")["full_candidate_adaptive_modes_agree_with_exact_flat_search"]
    ok, col = Collection.new(dimensions=4, metric=c["metric"], index="flat")
    assert col.put_many([{"id": r[0], "vector": r[1]} for r in c["rows"]]) == "ok"
    q, n, k = c["query"], c["candidates"], c["limit"]
    ids = [r.id for r in col.search(q, {"limit": k})[1]]
    gens = [("funnel", {"stages": [2, 4], "candidates": n}), ("quantized", {"candidates": n}), ("search", {"candidates": n})]
    ok, hybrid = col.hybrid_search(q, {"generators": gens, "limit": k})
    assert ok == "ok" and [r.id for r in hybrid] == ids
    assert [r.id for r in FlatGpu.hybrid_search(col, q, {"generators": gens, "limit": k})[1]] == ids
    assert col.hybrid_search(q, {"limit": 3})[0] == "ok" and col.hybrid_search(q, {"generators": ["search"], "limit": 3})[0] == "ok"
    assert col.hybrid_search(q, {"generators": []}) == ("error", "invalid_generators")
    assert col.hybrid_search(q, {"generators": ["nope"]}) == ("error", ("unknown_generator", "nope"))
    assert col.hybrid_search(q, {"generators": ["hnsw"]}) == ("error", "hnsw_index_required")
    assert col.hybrid_search(q, {"generators": [("funnel", {"stages": [5]})]}) == ("error", "invalid_stages")
    assert col.hybrid_search(q, {"generators": [("quantized", {"stages": [2]})]}) == ("error", ("unsupported_option", "stages"))
    assert col.hybrid_search(q, {"rerank": "nope"}) == ("error", ("invalid_rerank", "nope"))
    assert col.hybrid_search(q, {"limit": 0}) == ("error", "invalid_limit")
    assert col.hybrid_search(q, {"candidates": 5}) == ("error", ("unsupported_option", "candidates"))
    triples = lambda rs: [(r.id, r.score, r.distance) for r in rs]
    queries = [q, [v / 2 for v in q], [0.0, 0.0, 0.0, 0.0]]
    ok, qb = col.quantized_search_batch(queries, {"candidates": n, "limit": k})
    ok2, fb = col.funnel_search_batch(queries, {"stages": [2, 4], "candidates": n, "limit": k})
    ok3, sb = col.search_batch(queries, {"limit": k})
    assert (ok, ok2, ok3) == ("ok", "ok", "ok")
    for i, qi in enumerate(queries):
        assert triples(qb[i]) == triples(col.quantized_search(qi, {"candidates": n, "limit": k})[1])
        assert triples(fb[i]) == triples(col.funnel_search(qi, {"stages": [2, 4], "candidates": n, "limit": k})[1])
        assert triples(sb[i]) == triples(col.search(qi, {"limit": k})[1])
    assert [r.id for r in qb[0]] == ids and [r.id for r in fb[0]] == ids
    assert col.quantized_search_batch(queries, {"candidates": 3, "limit": 10}) == ("error", "invalid_candidates")
    assert col.funnel_search_batch([[1.0]], {"limit": 1}) == ("error", "dimension_mismatch")
    assert col.quantized_search(q, {"stages": [2]}) == ("error", ("unsupported_option", "stages"))


@pytest.mark.parametrize("metric", [2, 0])
def test_hybrid_search_matches_oracle_composition(nifs, oracle_mod, metric):
    """collection.ex:325-345, :515-592: union of generator candidates (funnel stages,
    sign-bit Hamming, the index's own search), exact rerank -- composed from oracle pieces."""
    n, d = 5000, 96
    x, ids = make_corpus(n, d, 410 + metric, metric == 2, oracle_mod, tie_block=25)
    packed = oracle_mod.pack_ids(ids)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rows = [(ids[i], x[i]) for i in range(n)]
    by_id = dict(rows)
    obits = [(ids[i], oracle_mod.compress_sign_bits(x[i])) for i in range(n)]
    rng = np.random.default_rng(6)
    for _ in range(3):
        q = rng.uniform(-1, 1, d).astype(np.float32)
        if metric == 2:
            q = oracle_mod.normalize_l2(q)
        cur = rows
        for st in (16, 48):
            kept = oracle_mod.vector_top_k(cur, q, metric, st, 60)
            cur = [(i, by_id[i]) for i, _ in kept]
        union = [i for i, _ in cur]
        union += [i for i, _ in oracle_mod.binary_top_k(obits, oracle_mod.compress_sign_bits(q), d, 80)]
        union += [i for i, _ in oracle_mod.matrix_search(metric, x, packed, q, 30)]
        uniq = list(dict.fromkeys(union))
        want = oracle_mod.vector_top_k([(i, by_id[i]) for i in uniq], q, metric, d, 10)
        got = unwrap(nifs.flat_hybrid_search(g.ref, q, [(nifs.GEN_FUNNEL, 60, [16, 48]), (nifs.GEN_QUANTIZED, 80, []),
                                                         (nifs.GEN_SEARCH, 30, [])], 10))
        assert bits(got) == bits(want), metric


@pytest.mark.parametrize("metric", [2, 0, 1, 3, 5, 6, 7, 8])
def test_hybrid_generator_mixes_match_oracle_composition(nifs, oracle_mod, metric):
    """hybrid_search (collection.ex:325-345, :515-592) for every metric, generator mix and overlap -- repeated generators
    (all repeats), a generator of one candidate, eight generators, funnels of one and of three stages -- on a corpus large
    enough for the histogram Hamming pass and on a small one: the union of the generators' candidates in order of first
    appearance, exact rerank, composed here from the oracle's pieces.  (Through r05 this test compared the host-composed
    path with a one-chain device form of it, VT_HYBRID_CHAIN; that form measured no faster and has left the library.)"""
    for n, d in ((20_000, 64), (900, 40)):
        x, ids = make_corpus(n, d, 470 + metric, metric == 2, oracle_mod, tie_block=25)
        packed = oracle_mod.pack_ids(ids)
        g = GpuIndex(nifs, metric)
        unwrap(nifs.flat_load_matrix(g.ref, ids, x))
        rows = [(ids[i], x[i]) for i in range(n)]
        by_id = dict(rows)
        obits = [(ids[i], oracle_mod.compress_sign_bits(x[i])) for i in range(n)]
        rng = np.random.default_rng(16 + metric)
        mixes = [
            [(nifs.GEN_FUNNEL, 60, [d // 4, d // 2]), (nifs.GEN_QUANTIZED, 80, []), (nifs.GEN_SEARCH, 30, [])],
            [(nifs.GEN_SEARCH, 256, [])],
            [(nifs.GEN_QUANTIZED, 200, []), (nifs.GEN_QUANTIZED, 200, [])],       # the same rows twice: all repeats
            [(nifs.GEN_FUNNEL, 256, [d]), (nifs.GEN_SEARCH, 1, []), (nifs.GEN_FUNNEL, 7, [3, 5, d])],
            [(nifs.GEN_SEARCH, 100, [])] * 8,
            [(nifs.GEN_SEARCH, 300, []), (nifs.GEN_QUANTIZED, 300, [])],          # beyond one fused list
        ]
        for gens in mixes:
            for limit in (1, 10, 256):
                q = rng.uniform(-1, 1, d).astype(np.float32)
                if metric == 2:
                    q = oracle_mod.normalize_l2(q)
                q[:2] = x[n // 2][:2]
                union = []
                for kind, cand, stages in gens:
                    if kind == nifs.GEN_FUNNEL:
                        cur = rows
                        for st in stages:
                            cur = [(i, by_id[i]) for i, _ in oracle_mod.vector_top_k(cur, q, metric, st, cand)]
                        union += [i for i, _ in cur]
                    elif kind == nifs.GEN_QUANTIZED:
                        union += [i for i, _ in oracle_mod.binary_top_k(obits, oracle_mod.compress_sign_bits(q), d, cand)]
                    else:
                        union += [i for i, _ in oracle_mod.matrix_search(metric, x, packed, q, cand)]
                uniq = list(dict.fromkeys(union))
                want = oracle_mod.vector_top_k([(i, by_id[i]) for i in uniq], q, metric, d, limit)
                got = unwrap(nifs.flat_hybrid_search(g.ref, q, gens, limit))
                assert bits(got) == bits(want), (metric, n, gens, limit)


def test_binary_top_k_with_massive_ties(nifs, oracle_mod):
    """One-word codes: only 65 distinct distances over 200k rows, so the k-th key
    sits in a crowded radix bin and ties are decided by id bytes alone."""
    rng = np.random.default_rng(64)
    n, d = 200_000, 64
    words = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    ids = [b"r%d" % i for i in range(n)]
    vecs = [(ids[i], [int(words[i])]) for i in range(n)]
    q = [int(words[123])]
    for limit in (10, 100, 700):
        got = unwrap(nifs.binary_top_k(vecs, q, d, limit))
        want = oracle_mod.binary_top_k(vecs, q, d, limit)
        assert got == want, limit


NOMINATE = {"f32": 1, "bf16": 2}   # VT_NOMINATE_*: which matrix-core pass names the candidates


@pytest.mark.parametrize("nominate", ["bf16", "f32"])
@pytest.mark.parametrize("metric", [2, 3, 4, 0, 1, 5])
def test_batched_search_equals_single_queries(nifs, oracle_mod, metric, nominate, monkeypatch, vt_debug):
    """vt_flat_search_batch: dot-family metrics go through a matrix-core candidate pass (K2b:
    operands rounded to bf16, the default; K2: FP32 matrix cores) + exact rescoring; every query
    must still equal the oracle bit for bit under both (BASELINE.json configs[2] shape, scaled
    down).  L2 / L2^2 nominate by 2 q.x - |x|^2; manhattan has no GEMM form and takes the
    per-query path."""
    vt_debug.set("force_batch_mfma", 1)   # the cost model would send these small corpora to single scans
    n, d = 20000, 192
    x, ids = make_corpus(n, d, 500 + metric, metric == 2, oracle_mod, tie_block=48)
    packed = oracle_mod.pack_ids(ids)
    g = GpuIndex(nifs, metric)
    assert nifs.flat_set_batch_nominate(g.ref, NOMINATE[nominate]) == "ok"
    assert nifs.flat_batch_nominate(g.ref) == NOMINATE[nominate]
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(8)
    nifs.flat_set_profiling(g.ref, True)
    # (600 and 4 096: three and sixteen groups of 256 in one call -- consecutive groups alternate between two contexts,
    # group g + 1 queued before group g is waited for; SURVEY 8d writes config 3 as "16 batches x 256")
    for nq, k in ((8, 10), (37, 1), (100, 10), (128, 3), (256, 10), (300, 64), (600, 10), (4096, 10)):
        qs = rng.uniform(-1, 1, size=(nq, d)).astype(np.float32)
        qs[0] = x[n // 2]  # sits on the block of identical rows
        if metric == 2:
            qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
        got = unwrap(nifs.flat_search_batch(g.ref, qs, k))
        assert len(got) == nq
        # (the largest batch: the first and last query of every group and every fifth in between against the oracle)
        check = range(nq) if nq <= 600 else sorted(set(range(0, nq, 5)) | set(range(255, nq, 256)) | set(range(0, nq, 256)))
        for i in check:
            assert bits(got[i]) == bits(oracle_mod.matrix_search(metric, x, packed, qs[i], k)), (metric, nq, k, i)
    prof = nifs.flat_get_profile(g.ref)
    key = "nominate" if nominate == "bf16" else "batch"
    other = "batch" if nominate == "bf16" else "nominate"
    assert prof[other + "_launches"] == 0, prof
    if metric != 5:
        assert prof[key + "_launches"] >= 5 and prof["batch_fallbacks"] <= prof[key + "_queries"] // 10, prof
    else:
        assert prof[key + "_launches"] == 0
    # validation order and empty cases follow flat_search
    assert nifs.flat_search_batch(g.ref, np.zeros((3, d + 1), np.float32), 5) == ("error", "dimension mismatch")
    bad = np.zeros((9, d), np.float32)
    bad[4, 3] = np.inf
    assert nifs.flat_search_batch(g.ref, bad, 5) == ("error", "vector contains a non-finite value")
    assert unwrap(nifs.flat_search_batch(g.ref, bad, 0)) == [[]] * 9


@pytest.mark.parametrize("nominate", ["bf16", "f32"])
def test_batched_search_large_values_fall_back_safely(nifs, oracle_mod, nominate, monkeypatch, vt_debug):
    """Huge coordinates blow the error margin (or overflow the MFMA sum): the
    bound must refuse and the per-query path must still give the exact answer."""
    vt_debug.set("force_batch_mfma", 1)
    scale = 1e18
    n, d = 6000, 64
    rng = np.random.default_rng(21)
    x = (rng.uniform(-1, 1, size=(n, d)) * scale).astype(np.float32)
    ids = [b"r%d" % i for i in range(n)]
    g = GpuIndex(nifs, 3)
    assert nifs.flat_set_batch_nominate(g.ref, NOMINATE[nominate]) == "ok"
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    qs = (rng.uniform(-1, 1, size=(16, d)) * scale).astype(np.float32)
    got = unwrap(nifs.flat_search_batch(g.ref, qs, 5))
    packed = oracle_mod.pack_ids(ids)
    for i in range(16):
        assert bits(got[i]) == bits(oracle_mod.matrix_search(3, x, packed, qs[i], 5)), i


@pytest.mark.parametrize("nominate", ["bf16", "f32"])
def test_rows_that_round_to_infinity_in_bf16_are_not_lost(nifs, oracle_mod, nominate, monkeypatch, vt_debug):
    """A coordinate at f32's largest value rounds to +inf in bf16; against a query that is zero
    there the exact dot product is finite (and these rows are the best hits), the bf16 one is
    inf * 0 = NaN and nominates nothing.  The handle knows its largest row norm: such a corpus
    is never certified by K2b, the exact paths answer."""
    vt_debug.set("force_batch_mfma", 1)
    n, d = 6000, 64
    rng = np.random.default_rng(22)
    x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    big = rng.choice(n, 12, replace=False)
    x[big, 5] = np.float32(3.4028235e38)
    x[big, 6] = 50.0                      # what makes them the best hits
    ids = [b"r%d" % i for i in range(n)]
    g = GpuIndex(nifs, 3)
    assert nifs.flat_set_batch_nominate(g.ref, NOMINATE[nominate]) == "ok"
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    qs = rng.uniform(-1, 1, size=(16, d)).astype(np.float32)
    qs[:, 5] = 0.0
    qs[:, 6] = 1.0
    got = unwrap(nifs.flat_search_batch(g.ref, qs, 5))
    packed = oracle_mod.pack_ids(ids)
    for i in range(16):
        want = oracle_mod.matrix_search(3, x, packed, qs[i], 5)
        assert bits(got[i]) == bits(want), i
        assert {h[0] for h in got[i]} <= {ids[j] for j in big}


@pytest.mark.parametrize("metric", [2, 3, 0])
def test_bf16_nomination_second_pass(nifs, oracle_mod, metric, monkeypatch, vt_debug):
    """K2b with a threshold that leaves no margin (VT_BF16_RANK = limit: tau is the k-th best
    bf16 score itself): the bound cannot certify anything in the first pass, every query names
    the threshold its k exact hits DO clear, and one more pass with those certifies them all --
    same hits as the oracle, no query left to the single-query path."""
    vt_debug.set("force_batch_mfma", 1)
    vt_debug.set("bf16_rank", 10)
    n, d = 30000, 256
    x, ids = make_corpus(n, d, 900 + metric, metric == 2, oracle_mod, tie_block=0)
    packed = oracle_mod.pack_ids(ids)
    g = GpuIndex(nifs, metric)
    assert nifs.flat_set_batch_nominate(g.ref, NOMINATE["bf16"]) == "ok"
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(77)
    qs = rng.uniform(-1, 1, size=(40, d)).astype(np.float32)
    if metric == 2:
        qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
    nifs.flat_set_profiling(g.ref, True)
    got = unwrap(nifs.flat_search_batch(g.ref, qs, 10))
    for i in range(40):
        assert bits(got[i]) == bits(oracle_mod.matrix_search(metric, x, packed, qs[i], 10)), (metric, i)
    prof = nifs.flat_get_profile(g.ref)
    assert prof["nominate_launches"] == 2 and prof["nominate_second_passes"] == 1, prof
    assert prof["batch_fallbacks"] == 0, prof


@pytest.mark.parametrize("metric,world", [(2, 4), (0, 4), (2, 8), (0, 8)])
def test_device_side_shard_merge_equals_single_index(nifs, oracle_mod, metric, world):
    """The multi-GPU exchange path on one GPU: 4 even / 8 UNEVEN row-block shards (the width of the node: one of them
    holds five rows, fewer than a list is long) whose id_rank columns are slices of ONE ordering of all ids
    (vt_rank_ids), per-shard vt_flat_search_begin into a gathered device buffer, vt_flat_merge_gathered --
    must equal the oracle over all rows, ties (identical rows across shards)
    included."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # the runtime the library already loaded (torch must not come second)
    n, d = 40_000, 96
    x, ids = make_corpus(n, d, 900 + metric, metric == 2, oracle_mod, tie_block=0)
    cuts = [s * (n // world) for s in range(world)] + [n] if world == 4 else [0, 3000, 3005, 11000, 12500, 21000, 30500, 39000, n]
    # identical rows living in different shards: only the global id order can rank them
    for s in range(world):
        x[cuts[s] + min(17, cuts[s + 1] - cuts[s] - 1)] = x[5]
    packed = oracle_mod.pack_ids(ids)
    ranks = nifs.rank_ids(nifs.pack_ids(ids))
    shards = []
    for s in range(world):
        g = GpuIndex(nifs, metric)
        unwrap(nifs.flat_load_matrix(g.ref, ids[cuts[s]:cuts[s + 1]], x[cuts[s]:cuts[s + 1]]))
        assert nifs.flat_set_id_ranks(g.ref, ranks[cuts[s]:cuts[s + 1]]) == "ok"
        shards.append(g)
    rng = np.random.default_rng(12)
    bufs = nifs.MergeBuffers()
    for limit in (10, 64):
        block_bytes = 16 + limit * 16
        gathered = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(gathered), world * block_bytes) == 0
        assert hip.hipMemset(gathered, 0, world * block_bytes) == 0
        for qi in range(4):
            q = x[5] if qi == 0 else rng.uniform(-1, 1, d).astype(np.float32)
            if metric == 2 and qi:
                q = oracle_mod.normalize_l2(q)
            for s, g in enumerate(shards):
                assert nifs.flat_search_begin(g.ref, q, limit, gathered.value + s * block_bytes) == "ok"
            assert hip.hipDeviceSynchronize() == 0
            st, cnt = nifs.flat_merge_gathered(shards[0].ref, gathered.value, world, limit, block_bytes, bufs)
            assert st == "ok" and cnt == limit
            got = [(ids[cuts[int(bufs.shard[i])] + int(bufs.rows[i])], float(bufs.raw[i])) for i in range(cnt)]
            assert bits(got) == bits(oracle_mod.matrix_search(metric, x, packed, q, limit)), (metric, limit, qi)
        assert hip.hipFree(gathered) == 0
    # a mutation invalidates the external ranks; the shard re-ranks locally and still answers alone
    shards[0].insert("zzz-new", x[0])
    assert shards[0].search(x[0], 2)[0][1] == shards[0].search(x[0], 2)[1][1]


# --------------------------------------------- full-size checks (BASELINE sizes)
def test_config2_full_size_properties_and_spot_parity(nifs, oracle_mod):
    """BASELINE.json configs[1]: flat cosine, d=768, N=1M, single query.
    Spot parity against the oracle on 2 queries (about 1 s of CPU each) plus
    size-independent properties on more."""
    n, d = 1_000_000, 768
    rng = np.random.default_rng(20260721)
    x = rng.uniform(-1.0, 1.0, size=(n, d)).astype(np.float32)
    x /= np.sqrt(np.sum(x.astype(np.float64) ** 2, axis=1, keepdims=True)).astype(np.float32)
    dup = rng.integers(0, n, n // 100)
    x[rng.integers(0, n, n // 100)] = x[dup]
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    packed = oracle_mod.pack_ids(ids)
    g = GpuIndex(nifs, 2)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    assert len(g) == n
    qrng = np.random.default_rng(20260722)
    for i in range(2):
        q = oracle_mod.normalize_l2(qrng.uniform(-1, 1, d).astype(np.float32))
        assert bits(g.search(q, 10)) == bits(oracle_mod.matrix_search(2, x, packed, q, 10)), i
    id_to_row = None
    for i in range(8):
        row = int(qrng.integers(0, n))
        hits = g.search(x[row], 10)
        # idempotence / self-hit: a stored row is its own best match (or ties with its verbatim duplicates)
        top = hits[0]
        assert close(top[1], 1.0, 1e-5)
        # sortedness under the reference order: (rank = 1 - raw in f32, id bytes)
        keys = [(support.total_key(np.float32(1.0) - np.float32(h[1])), h[0]) for h in hits]
        assert keys == sorted(keys)
        # limit monotonicity: top-5 is a prefix of top-10
        assert bits(g.search(x[row], 5)) == bits(hits[:5])


def test_concurrent_callers_on_shared_and_separate_handles(nifs, oracle_mod):
    """nifs.rs:266-309: every NIF is a dirty-scheduler job, so several OS threads call into
    the same and into different indexes at once (searches under the handle's lock, a writer
    mutating a third index).  ctypes releases the GIL during the C call; every result must
    still equal the oracle's."""
    import threading
    specs = [(2, 6000, 64, True), (0, 5000, 100, False)]
    idx, want = [], []
    for metric, n, d, norm in specs:
        x, ids = make_corpus(n, d, 900 + metric, norm, oracle_mod, tie_block=20)
        g = GpuIndex(nifs, metric)
        unwrap(nifs.flat_load_matrix(g.ref, ids, x))
        packed = oracle_mod.pack_ids(ids)
        rng = np.random.default_rng(31 + metric)
        qs = [x[n // 2]] + [rng.uniform(-1, 1, d).astype(np.float32) for _ in range(5)]
        if norm:
            qs = [oracle_mod.normalize_l2(q) for q in qs]
        idx.append(g)
        want.append([(q, bits(oracle_mod.matrix_search(metric, x, packed, q, 10))) for q in qs])
    # a third index that a writer thread keeps mutating
    wx, wids = make_corpus(3000, 32, 977, False, oracle_mod)
    writer = GpuIndex(nifs, 1)
    unwrap(nifs.flat_load_matrix(writer.ref, wids[:2000], wx[:2000]))
    errors = []

    def reader(t):
        try:
            for i in range(150):
                which = (t + i) % 2
                q, expect = want[which][(t * 7 + i) % len(want[which])]
                got = bits(unwrap(nifs.flat_search(idx[which].ref, q, 10)))
                if got != expect:
                    errors.append(("reader", t, i, which))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(("reader", t, repr(e)))

    def mutate():
        try:
            for i in range(2000, 3000):
                writer.insert(wids[i], wx[i])
                if i % 3 == 0:
                    writer.delete(wids[i - 1500])
                if i % 50 == 0:
                    writer.search(wx[i], 5)
        except Exception as e:  # noqa: BLE001
            errors.append(("writer", repr(e)))

    threads = [threading.Thread(target=reader, args=(t,)) for t in range(6)] + [threading.Thread(target=mutate)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    # the mutated index still answers like the oracle index that saw the same operations
    o = oracle_mod.FlatIndex(1)
    o.insert_matrix(wids[:2000], wx[:2000])
    for i in range(2000, 3000):
        o.insert(wids[i], wx[i])
        if i % 3 == 0:
            o.delete(wids[i - 1500])
    assert len(writer) == len(o)
    for q in (wx[10], wx[2500], wx[2999]):
        assert bits(writer.search(q, 10)) == bits(o.search(q, 10))


def test_quantized_histogram_pass_survives_dimension_changes(nifs, oracle_mod):
    """An emptied index forgets its dimension (flat.rs:88-93); refilled with another one, the
    histogram pass must not see counts left over from the wider rows."""
    rng = np.random.default_rng(99)
    g = GpuIndex(nifs, 0)
    n = 17_000
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    for d in (128, 32, 128):
        x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
        unwrap(nifs.flat_load_matrix(g.ref, ids, x))
        for _ in range(2):
            q = rng.uniform(-1, 1, d).astype(np.float32)
            ham = ((x >= 0) != (q >= 0)[None, :]).sum(axis=1)
            order = sorted(range(n), key=lambda i: (int(ham[i]), ids[i]))[:50]
            want = oracle_mod.vector_top_k([(ids[i], x[i]) for i in order], q, 0, d, 10)
            assert bits(unwrap(nifs.flat_quantized_search(g.ref, q, 50, 10))) == bits(want), d
        for i in ids:
            g.delete(i)
        assert len(g) == 0 and g.dimension is None


def test_padding_columns_of_a_small_batch_nominate_nothing(nifs, oracle_mod, monkeypatch, vt_debug):
    """A batch of 8 is padded to 32 query columns; the all-zero padding columns once passed
    every row as a candidate.  Parity of the small batch here; the timing guard that caught
    it lives in tests/test_gpu_perf.py (-m gpu_perf)."""
    vt_debug.set("force_batch_mfma", 1)
    n, d = 300_000, 128
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    g = GpuIndex(nifs, 3)
    assert nifs.flat_set_batch_nominate(g.ref, NOMINATE["f32"]) == "ok"   # (K2b always carries 256 columns: 248 padding ones)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    qs = rng.uniform(-1, 1, (32, d)).astype(np.float32)
    out8 = unwrap(nifs.flat_search_batch(g.ref, qs[:8], 10))
    for i in range(8):
        assert bits(out8[i]) == bits(unwrap(nifs.flat_search(g.ref, qs[i], 10)))
    assert nifs.flat_set_batch_nominate(g.ref, NOMINATE["bf16"]) == "ok"
    out8 = unwrap(nifs.flat_search_batch(g.ref, qs[:8], 10))
    for i in range(8):
        assert bits(out8[i]) == bits(unwrap(nifs.flat_search(g.ref, qs[i], 10)))


@pytest.fixture
def force_threshold(monkeypatch, vt_debug):
    """The library picks the threshold path by a cost model (large corpora); the parity tests
    force it so that it runs at oracle-sized inputs too.  getenv is read per call."""
    vt_debug.set("force_threshold_select", 1)


@pytest.mark.parametrize("forced", [True, False])
@pytest.mark.parametrize("metric", [2, 0, 5])
def test_limits_above_256_in_one_scan(nifs, oracle_mod, metric, forced, monkeypatch, vt_debug):
    """Above 65 536 rows a limit of 257..4096 is answered from a key column and a radix
    threshold instead of one scan per 256 hits; 5 000 still takes the pass-per-256 loop.
    Both must equal the oracle's full sort, ties by id bytes included."""
    if forced:
        vt_debug.set("force_threshold_select", 1)
    n, d = 70_000, 24
    x, ids = make_corpus(n, d, 1200 + metric, metric == 2, oracle_mod, tie_block=600)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    packed = oracle_mod.pack_ids(ids)
    rng = np.random.default_rng(8)
    for limit in (300, 1000, 4096, 5000):
        q = x[n // 2] if limit == 1000 else rng.uniform(-1, 1, d).astype(np.float32)   # 1000: inside the tie block
        if metric == 2:
            q = oracle_mod.normalize_l2(q)
        want = oracle_mod.matrix_search(metric, x, packed, q, limit)
        got = unwrap(nifs.flat_search(g.ref, q, limit))
        assert bits(got) == bits(want), (metric, limit)


@pytest.mark.parametrize("metric", [2, 0])
def test_limits_above_4096_in_one_scan(nifs, oracle_mod, metric):
    """flat.ex:98-103 allows any limit below 2^32.  Above 4 096 hits the scan writes key and
    payload columns, an exact radix threshold (all 64 key bits) runs on the device and the host
    orders the k collected entries: ONE scan whatever the limit (the profile counts the
    launches) -- 70 000 of 90 000 rows, all of them, more than there are, usize::MAX."""
    n, d = 90_000, 24
    x, ids = make_corpus(n, d, 1500 + metric, metric == 2, oracle_mod, tie_block=700)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    packed = oracle_mod.pack_ids(ids)
    rng = np.random.default_rng(18)
    nifs.flat_set_profiling(g.ref, True)
    for limit, scans in ((5000, 1), (20000, 1), (65536, 1), (70000, 1), (89_999, 1), (n, 1), (n + 5, 1), (2 ** 64 - 1, 1)):
        q = x[n // 2] if limit == 20000 else rng.uniform(-1, 1, d).astype(np.float32)
        if metric == 2:
            q = oracle_mod.normalize_l2(q)
        nifs.flat_get_profile(g.ref, reset=True)
        got = unwrap(nifs.flat_search(g.ref, q, limit))
        prof = nifs.flat_get_profile(g.ref, reset=True)
        assert bits(got) == bits(oracle_mod.matrix_search(metric, x, packed, q, limit)), (metric, limit)
        if scans is not None:
            assert prof["scan_launches"] == scans, (limit, prof["scan_launches"])


def test_limit_above_256_with_massive_ties(nifs, oracle_mod, force_threshold):
    """80 000 identical rows: every key shares its rank (r02: the 33-bit threshold collected them
    all, the device list overflowed, the call fell back to one scan per 256 hits).  The threshold
    now resolves all 64 bits -- the id ranks cut the tie -- and the answer comes from one scan."""
    n, d = 80_000, 16
    x = np.tile(np.random.default_rng(3).uniform(-1, 1, (1, d)).astype(np.float32), (n, 1))
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    q = np.zeros(d, np.float32)
    nifs.flat_set_profiling(g.ref, True)
    for limit in (400, 4096, 30_000):
        nifs.flat_get_profile(g.ref, reset=True)
        want = oracle_mod.matrix_search(0, x, oracle_mod.pack_ids(ids), q, limit)
        assert bits(unwrap(nifs.flat_search(g.ref, q, limit))) == bits(want)
        assert nifs.flat_get_profile(g.ref, reset=True)["scan_launches"] <= 2, limit   # (+1: the winners' raw values re-scored)


@pytest.mark.parametrize("metric", [7, 8])
def test_few_distinct_values_at_limit_1000(nifs, oracle_mod, metric, force_threshold):
    """Float hamming / jaccard take a handful of distinct values (limit 1000
    cost 1.15 ms where the other metrics took 0.28 -- the tie list overflowed): one scan now."""
    n, d = 70_000, 24
    rng = np.random.default_rng(40 + metric)
    x = (rng.uniform(-1, 1, (n, d)) * (rng.uniform(0, 1, (n, d)) < 0.5)).astype(np.float32)
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    q = x[7]
    nifs.flat_set_profiling(g.ref, True)
    nifs.flat_get_profile(g.ref, reset=True)
    got = unwrap(nifs.flat_search(g.ref, q, 1000))
    assert nifs.flat_get_profile(g.ref, reset=True)["scan_launches"] <= 2
    assert bits(got) == bits(oracle_mod.matrix_search(metric, x, oracle_mod.pack_ids(ids), q, 1000))


@pytest.mark.parametrize("metric", [2, 0])
def test_funnel_with_more_than_256_candidates(nifs, oracle_mod, metric, force_threshold):
    """funnel_search(limit: 100) defaults to 1 000 candidates (collection.ex:547): the first stage
    over all rows keeps them through the key-column threshold."""
    n, d = 70_000, 48
    x, ids = make_corpus(n, d, 1300 + metric, metric == 2, oracle_mod, tie_block=30)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rows = [(ids[i], x[i]) for i in range(n)]
    by_id = dict(rows)
    rng = np.random.default_rng(4)
    for stages, cand, limit in (([16], 1000, 100), ([8, 24], 600, 20)):
        q = rng.uniform(-1, 1, d).astype(np.float32)
        if metric == 2:
            q = oracle_mod.normalize_l2(q)
        cur = rows
        for st in stages:
            kept = oracle_mod.vector_top_k(cur, q, metric, st, cand)
            cur = [(i, by_id[i]) for i, _ in kept]
        want = oracle_mod.vector_top_k(cur, q, metric, d, limit)
        got = unwrap(nifs.flat_funnel_search(g.ref, q, stages, cand, limit))
        assert bits(got) == bits(want), (metric, stages, cand, limit)


def test_limit_above_256_with_an_overflowing_row_reports_the_error(nifs, oracle_mod, force_threshold):
    """A row whose score overflows leaves an excluded slot in the key column; the threshold path
    must still end in "metric overflow" (distances.rs:67), not in a short or garbage list."""
    n, d = 70_000, 16
    x = np.random.default_rng(2).uniform(-1, 1, (n, d)).astype(np.float32)
    x[12345, :] = 3e38
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    g = GpuIndex(nifs, 3)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    q = np.full(d, 2.0, np.float32)
    assert nifs.flat_search(g.ref, q, 500) == ("error", "metric overflow")
    assert nifs.flat_search(g.ref, np.zeros(d, np.float32), 500)[0] == "ok"   # and the flag does not stick


def test_searches_between_unsorted_inserts_stay_exact(nifs, oracle_mod):
    """Ids arriving out of bytewise order leave their rows unranked until a search needs the true
    id order: a search first tries with one extra hit and re-ranks only if the boundary ties.
    Interleaved inserts (with verbatim copies of stored rows under smaller and larger ids, so that
    ties fall inside the list and across its boundary), deletes and searches against the oracle."""
    d = 12
    rng = np.random.default_rng(77)
    base = rng.uniform(-1, 1, (400, d)).astype(np.float32)
    for metric in (0, 2, 3):
        g = GpuIndex(nifs, metric)
        o = oracle_mod.FlatIndex(metric)
        rows = {}

        def put(id_, v):
            g.insert(id_, v)
            o.insert(id_, v)
            rows[id_] = v

        for i in range(300):
            put(b"m-%04d" % i, base[i])
        step = 0
        for i in range(300, 400):
            # alternate: new vector under an id that sorts before / after everything; copies of stored rows
            kind = i % 4
            if kind == 0:
                put(b"a-%d" % i, base[i])
            elif kind == 1:
                put(b"z-%d" % i, base[i])
            elif kind == 2:
                put(b"a-copy-%d" % i, base[i - 250])     # ties with m-(i-250), smaller id
            else:
                put(b"zz-copy-%d" % i, base[i - 250])    # ties, larger id
            if i % 7 == 0:
                victim = b"m-%04d" % (i - 290)
                g.delete(victim)
                o.delete(victim)
                rows.pop(victim, None)
            for q in (base[i - 250], base[(i * 3) % 300], rng.uniform(-1, 1, d).astype(np.float32)):
                for k in (1, 2, 5, 255, 256):
                    step += 1
                    assert bits(g.search(q, k)) == bits(o.search(q, k)), (metric, i, k)
        assert len(g) == len(o)


def test_equal_keys_among_unranked_rows(nifs, oracle_mod):
    """Rows inserted out of id order share one sentinel rank until the next re-rank, so two of them
    at the same f32 distance carry the SAME 64-bit key.  A top-k compaction that keeps "the first k
    keys <= threshold" then drops a smaller key when the threshold has equals in front of it -- r01's
    bug: `limit: 1` returned the second-best row (1 in ~1e6 searches on random data, found by a soak
    test).  Here deterministically: the duplicates sit in lower lanes than the true best row."""
    d = 8
    far = np.full(d, 4.0, np.float32)
    a = np.zeros(d, np.float32); a[0] = 2.0
    q = np.zeros(d, np.float32)
    rows = [("z9", far), ("y8", a), ("x7", a), ("w6", q)] + \
           [("v%02d" % (40 - i), np.full(d, 3.0 + i / 64, np.float32)) for i in range(36)]
    for metric in (0, 1, 5):
        g = GpuIndex(nifs, metric)
        o = oracle_mod.FlatIndex(metric)
        for id_, v in rows:                     # one by one, ids descending: every row but the first is unranked
            g.insert(id_, v)
            o.insert(id_, v)
        for k in (1, 2, 3, 4, 10):
            assert bits(g.search(q, k)) == bits(o.search(q, k)), (metric, k)
        assert g.search(q, 1)[0][0] == b"w6"
    # the same at scale: coordinates from {-1, 0, 1} give thousands of exactly equal distances; ids in
    # shuffled order (all unranked), limits on both sides of the buffer sizes (long lists go through
    # the radix select, short ones through the counting one)
    rng = np.random.default_rng(12)
    n, d = 60_000, 12
    x = rng.integers(-1, 2, (n, d)).astype(np.float32)
    ids = [b"k%06d" % i for i in rng.permutation(n)]
    g = GpuIndex(nifs, 0)
    o = oracle_mod.FlatIndex(0)
    for s0 in range(0, n, 4000):               # below the bulk-ranking size: rows stay unranked
        g.insert_many([(ids[i], x[i]) for i in range(s0, s0 + 4000)])
    o.insert_matrix(ids, x)
    for k in (1, 2, 7, 30, 64, 65, 200, 255):
        q = rng.integers(-1, 2, d).astype(np.float32)
        assert bits(g.search(q, k)) == bits(o.search(q, k)), k


def test_derived_data_is_patched_after_mutations(nifs, oracle_mod, monkeypatch, vt_debug):
    """Sign bits (quantized_search) and row norms (batched L2) are kept per row and patched for the
    rows an insert / upsert / delete touched instead of being rebuilt; every search in between
    must still agree with the oracle."""
    vt_debug.set("force_batch_mfma", 1)
    n, d = 20_000, 64
    rng = np.random.default_rng(404)
    x = rng.uniform(-1, 1, (n + 200, d)).astype(np.float32)
    ids = [b"doc-%05d" % i for i in range(n + 200)]
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids[:n], x[:n]))
    cur = {ids[i]: x[i] for i in range(n)}

    def check():
        keys = sorted(cur)
        mat = np.stack([cur[k] for k in keys])
        packed = oracle_mod.pack_ids(keys)
        q = rng.uniform(-1, 1, d).astype(np.float32)
        ham = ((mat >= 0) != (q >= 0)[None, :]).sum(axis=1)
        order = sorted(range(len(keys)), key=lambda i: (int(ham[i]), keys[i]))[:40]
        want = oracle_mod.vector_top_k([(keys[i], mat[i]) for i in order], q, 0, d, 10)
        assert bits(unwrap(nifs.flat_quantized_search(g.ref, q, 40, 10))) == bits(want)
        qs = rng.uniform(-1, 1, (4, d)).astype(np.float32)
        got = unwrap(nifs.flat_search_batch(g.ref, qs, 5))
        for i in range(4):
            assert bits(got[i]) == bits(oracle_mod.matrix_search(0, mat, packed, qs[i], 5))

    check()                                              # builds bits and norms
    for step in range(6):
        for j in range(20):                              # appends, upserts (one with a huge norm), deletes
            i = n + step * 20 + j
            g.insert(ids[i], x[i]); cur[ids[i]] = x[i]
        up = ids[step * 37]
        v = x[step * 37] * (50.0 if step == 3 else -1.0)
        g.insert(up, v); cur[up] = v
        for victim in (ids[1000 + step], ids[n - 1 - step]):
            g.delete(victim); cur.pop(victim, None)
        check()


def test_boundary_ties_after_a_bulk_load(nifs):
    """After a bulk load (ids in numeric, not bytewise, order; verbatim duplicate rows, as in
    bench.py) limit-1 queries on planted copies tie across the boundary: the copy with the
    bytewise smaller id wins.  (The per-query timing guard of this scenario: test_gpu_perf.py.)"""
    import torch
    from bench import build_shard, doc_ids
    rows, dim = 1_000_000, 256
    x = build_shard(torch, torch.device("cuda", 0), rows, dim, 4242)
    x[500_000:500_032] = x[:32]
    g = GpuIndex(nifs, 2)
    assert nifs.flat_load_device_matrix(g.ref, doc_ids(0, rows), x.data_ptr(), rows, dim) == ("ok", ())
    dup = x[:32].cpu().numpy()
    del x
    for i, q in enumerate(dup):
        hits = unwrap(nifs.flat_search(g.ref, q, 1))
        assert hits[0][0] == min(b"doc-%d" % (i + 1), b"doc-%d" % (500_000 + i + 1))


# ------------------------------------------------- the surface at the reference's limits
@pytest.mark.parametrize("d", [4096, 5000, 9001])
def test_jaccard_beyond_4095_dimensions(nifs, oracle_mod, d):
    """distances.rs:327-347 has no bound on d.  The device packs (hamming count, non-zero count)
    into one f32 per panel and carries them as integers across panels (r02: d >= 4096 was
    VT_ERR_UNSUPPORTED)."""
    n = 600
    rng = np.random.default_rng(d)
    x = (rng.uniform(-1, 1, (n, d)) * (rng.uniform(0, 1, (n, d)) < 0.3)).astype(np.float32)
    x[5] = 0.0                                                       # an all-zero row: union with a zero query is empty
    ids = [b"j-%d" % i for i in range(n)]
    packed = oracle_mod.pack_ids(ids)
    for metric in (8, 7):
        g = GpuIndex(nifs, metric)
        unwrap(nifs.flat_load_matrix(g.ref, ids, x))
        for q in (x[17], np.zeros(d, np.float32), (rng.uniform(-1, 1, d) * (rng.uniform(0, 1, d) < 0.5)).astype(np.float32)):
            for limit in (1, 10, 300):
                assert bits(g.search(q, limit)) == bits(oracle_mod.matrix_search(metric, x, packed, q, limit)), (metric, d, limit)
        vecs = [(ids[i], x[i]) for i in range(50)]
        assert bits(unwrap(nifs.vector_top_k(vecs, x[3], metric, d, 7))) == bits(oracle_mod.vector_top_k(vecs, x[3], metric, d, 7))


@pytest.mark.parametrize("d", [24_000, 40_000])
def test_rows_too_long_for_the_query_to_sit_in_lds(nifs, oracle_mod, d):
    """flat.rs has no bound on the dimension.  Beyond ~24 000 floats the query no longer fits in
    LDS beside the scan's panels: the run-time-op kernel then reads its query fragments from
    global memory (r02: VT_ERR_UNSUPPORTED).  Every metric, bit for bit, plus batch and big limits."""
    n = 400
    rng = np.random.default_rng(d)
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    x[9] = x[3]
    ids = [b"w-%d" % i for i in range(n)]
    packed = oracle_mod.pack_ids(ids)
    q = rng.uniform(-1, 1, d).astype(np.float32)
    for metric in (0, 1, 2, 3, 4, 5, 6, 7, 8):
        g = GpuIndex(nifs, metric)
        unwrap(nifs.flat_load_matrix(g.ref, ids, x))
        for query, limit in ((q, 10), (x[3], 2), (q, n)):
            assert bits(g.search(query, limit)) == bits(oracle_mod.matrix_search(metric, x, packed, query, limit)), (metric, d, limit)
        if metric in (3, 5):
            got = unwrap(nifs.flat_search_batch(g.ref, np.stack([q, x[3], x[100]]), 5))
            for i, query in enumerate((q, x[3], x[100])):
                assert bits(got[i]) == bits(oracle_mod.matrix_search(metric, x, packed, query, 5))


# ------------------------------------------------- several quantized searches per sweep
def test_quantized_groups_with_partial_and_odd_words(nifs, oracle_mod):
    """Row lengths whose sign bits end in a partial word and / or an odd word count (the grouped
    pass relies on the padding bits and the pad word of rows and queries being zero)."""
    rng = np.random.default_rng(77)
    for d in (150, 65, 129, 200, 1):
        n = 17_000
        x, ids = make_corpus(n, d, 4300 + d, False, oracle_mod, tie_block=30)
        g = GpuIndex(nifs, 0)
        unwrap(nifs.flat_load_matrix(g.ref, ids, x))
        qs = rng.uniform(-1, 1, (8, d)).astype(np.float32)
        qs[1] = -np.abs(qs[1])            # every sign bit clear
        qs[2] = np.abs(qs[2])             # every sign bit set
        got = unwrap(nifs.flat_quantized_search_batch(g.ref, qs, 64, 10))
        sign = x >= 0
        for i in range(8):
            assert bits(got[i]) == bits(unwrap(nifs.flat_quantized_search(g.ref, qs[i], 64, 10))), (d, i)
            ham = (sign != (qs[i] >= 0)[None, :]).sum(axis=1)
            order = sorted(range(n), key=lambda r: (int(ham[r]), ids[r]))[:64]
            want = oracle_mod.vector_top_k([(ids[r], x[r]) for r in order], qs[i], 0, d, 10)
            assert bits(got[i]) == bits(want), (d, i)


@pytest.mark.parametrize("metric", [2, 0, 3, 5, 8])
def test_quantized_search_batch_equals_single_calls(nifs, oracle_mod, metric):
    """vt_flat_quantized_search_batch: groups of up to eight queries share ONE sweep of the sign
    bits (hamming_dist_multi_kernel), every later stage runs with the queries on grid.y; each
    query's hits must equal its own quantized_search AND the oracle's composition
    (binary_top_k then vector_top_k, collection.ex:276-295) bit for bit.  Jaccard's rerank takes
    the queries one by one (its non-zero count is per query)."""
    n, d = 20_000, 200
    x, ids = make_corpus(n, d, 4100 + metric, metric == 2, oracle_mod, tie_block=300)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(11)
    sign = x >= 0
    nifs.flat_set_profiling(g.ref, True)
    for nq, cand, limit in ((2, 100, 10), (8, 100, 10), (17, 256, 30), (5, 7, 7), (9, 40, 100)):
        qs = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
        qs[0] = x[n // 2]                                    # inside the block of identical rows
        if metric == 2:
            qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
        nifs.flat_get_profile(g.ref, reset=True)
        got = unwrap(nifs.flat_quantized_search_batch(g.ref, qs, cand, limit))
        prof = nifs.flat_get_profile(g.ref, reset=True)
        assert len(got) == nq
        if metric != 8:
            assert prof["hamming_queries"] >= nq - 1 and prof["hamming_launches"] <= (nq + 7) // 8 + 1, (nq, prof)
        for i in range(nq):
            assert bits(got[i]) == bits(unwrap(nifs.flat_quantized_search(g.ref, qs[i], cand, limit))), (metric, nq, i)
        for i in (0, nq - 1):
            ham = (sign != (qs[i] >= 0)[None, :]).sum(axis=1)
            order = sorted(range(n), key=lambda r: (int(ham[r]), ids[r]))[:cand]
            want = oracle_mod.vector_top_k([(ids[r], x[r]) for r in order], qs[i], metric, d, limit)
            assert bits(got[i]) == bits(want), (metric, nq, cand, limit, i)
    # validation order and empty cases follow quantized_search
    assert nifs.flat_quantized_search_batch(g.ref, np.zeros((3, d + 1), np.float32), 10, 5) == ("error", "dimension mismatch")
    bad = np.zeros((4, d), np.float32)
    bad[2, 1] = np.nan
    assert nifs.flat_quantized_search_batch(g.ref, bad, 10, 5) == ("error", "vector contains a non-finite value")
    assert unwrap(nifs.flat_quantized_search_batch(g.ref, qs[:3], 0, 5)) == [[]] * 3
    assert unwrap(nifs.flat_quantized_search_batch(g.ref, qs[:3], 5, 0)) == [[]] * 3


def test_funnel_search_batch_equals_single_calls(nifs, oracle_mod):
    """vt_flat_funnel_search_batch: on a cosine collection groups of up to eight queries share ONE
    sweep of the rows' prefixes (cosine_scan_multi_kernel: exact f64 cosines, a sampled threshold,
    per-query lists, batched select), later stages and the exact rerank run with the queries on
    grid.y; each query's hits must equal its own funnel_search AND the oracle's composition
    (vector_top_k on each prefix, then on the full vectors; collection.ex:245-260) bit for bit.
    The pattern metrics and large candidate counts take the calls one by one."""
    n, d = 30_000, 160
    x, ids = make_corpus(n, d, 4500, True, oracle_mod, tie_block=300)
    g = GpuIndex(nifs, 2)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rows = [(ids[i], x[i]) for i in range(n)]
    by_id = dict(rows)
    rng = np.random.default_rng(21)
    nifs.flat_set_profiling(g.ref, True)
    for nq, stages, cand, limit in ((2, [32], 100, 10), (8, [64, 128], 60, 10), (17, [16], 256, 30), (5, [160], 7, 7),
                                    (9, [8, 24, 96], 40, 100), (3, [1], 50, 5)):
        qs = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
        qs[0] = x[n // 2]                                    # inside the block of identical rows
        qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
        if nq >= 5:
            qs[3, :stages[0]] = 0.0                          # a zero prefix: every stage-1 cosine is 0
        nifs.flat_get_profile(g.ref, reset=True)
        got = unwrap(nifs.flat_funnel_search_batch(g.ref, qs, stages, cand, limit))
        prof = nifs.flat_get_profile(g.ref, reset=True)
        assert len(got) == nq
        assert prof["prefix_queries"] >= nq - 1 - (1 if nq >= 5 else 0), (nq, prof)   # (a lone last query goes alone; the zero prefix may)
        for i in range(nq):
            assert bits(got[i]) == bits(unwrap(nifs.flat_funnel_search(g.ref, qs[i], stages, cand, limit))), (nq, stages, i)
        for i in (0, nq - 1):
            cur = rows
            for st in stages:
                kept = oracle_mod.vector_top_k(cur, qs[i], 2, st, cand)
                cur = [(j, by_id[j]) for j, _ in kept]
            want = oracle_mod.vector_top_k(cur, qs[i], 2, d, limit)
            assert bits(got[i]) == bits(want), (nq, stages, cand, limit, i)
    # validation and empty cases follow funnel_search
    assert nifs.flat_funnel_search_batch(g.ref, qs[:3], [0], 10, 5) == ("error", "invalid prefix dimensions")
    assert nifs.flat_funnel_search_batch(g.ref, qs[:3], [d + 1], 10, 5) == ("error", "invalid prefix dimensions")
    assert nifs.flat_funnel_search_batch(g.ref, qs[:3], [], 10, 5) == ("error", "invalid prefix dimensions")
    assert nifs.flat_funnel_search_batch(g.ref, np.zeros((3, d + 1), np.float32), [4], 10, 5) == ("error", "dimension mismatch")
    assert unwrap(nifs.flat_funnel_search_batch(g.ref, qs[:3], [4], 0, 5)) == [[]] * 3
    assert unwrap(nifs.flat_funnel_search_batch(g.ref, qs[:3], [4], 5, 0)) == [[]] * 3
    # another metric (its own grouped sweep: test_funnel_batches_of_the_k1_families_share_the_prefix_sweep), same answers
    g0 = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g0.ref, ids, x))
    qs = rng.uniform(-1, 1, (4, d)).astype(np.float32)
    got = unwrap(nifs.flat_funnel_search_batch(g0.ref, qs, [32, 64], 50, 10))
    for i in range(4):
        assert bits(got[i]) == bits(unwrap(nifs.flat_funnel_search(g0.ref, qs[i], [32, 64], 50, 10)))


@pytest.mark.parametrize("metric", [0, 3, 1, 4, 5, 6])
def test_funnel_batches_of_the_k1_families_share_the_prefix_sweep(nifs, oracle_mod, metric):
    """vt_flat_funnel_search_batch on L2 / dot / L1 / Linf collections: groups of up to eight share ONE sweep of the
    prefixes (K1p, prefix_multi_kernel: K1's chunked f32 arithmetic per query; threshold from a sample on
    -rank_value; per-query lists cut to `candidates`), later stages and the rerank run K1's batch mode over each
    query's candidates.  Every query's hits equal its own funnel_search and the oracle's composition
    (collection.ex:245-260: vector_top_k on each prefix, then on the full rows) bit for bit -- prefixes with and
    without a scalar tail, narrower and wider than one 64-float panel, identical rows, all four lane orders."""
    n, d = 30_000, 200
    x, ids = make_corpus(n, d, 4700 + metric, False, oracle_mod, tie_block=300)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rows = [(ids[i], x[i]) for i in range(n)]
    by_id = dict(rows)
    rng = np.random.default_rng(31 + metric)
    nifs.flat_set_profiling(g.ref, True)
    shapes = ((2, [32], 100, 10), (8, [64, 128], 60, 10), (17, [13], 256, 30), (5, [200], 7, 7), (9, [8, 27, 96], 40, 100),
              (3, [1], 50, 5), (6, [70], 120, 12))
    if metric not in (0, 3):
        shapes = shapes[1:3] + shapes[6:]
    for nq, stages, cand, limit in shapes:
        qs = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
        qs[0] = x[n // 2]                                    # inside the block of identical rows
        nifs.flat_get_profile(g.ref, reset=True)
        got = unwrap(nifs.flat_funnel_search_batch(g.ref, qs, stages, cand, limit))
        prof = nifs.flat_get_profile(g.ref, reset=True)
        assert len(got) == nq
        assert prof["prefix_queries"] >= nq - 1 - nq // 8, (nq, stages, prof)   # (a lone last query goes alone; a threshold may miss)
        for i in range(nq):
            assert bits(got[i]) == bits(unwrap(nifs.flat_funnel_search(g.ref, qs[i], stages, cand, limit))), (metric, nq, stages, i)
        for i in (0, nq - 1):
            cur = rows
            for st in stages:
                kept = oracle_mod.vector_top_k(cur, qs[i], metric, st, cand)
                cur = [(j, by_id[j]) for j, _ in kept]
            want = oracle_mod.vector_top_k(cur, qs[i], metric, d, limit)
            assert bits(got[i]) == bits(want), (metric, nq, stages, cand, limit, i)
    if metric in (0, 3):
        # the other lane orders of wide's reduce_add: the sweep's arithmetic follows the handle's
        for order in (0, 1, 2):
            h = GpuIndex(nifs, metric, order=order)
            unwrap(nifs.flat_load_matrix(h.ref, ids[:20_000], x[:20_000]))
            qs = rng.uniform(-1, 1, (5, d)).astype(np.float32)
            got = unwrap(nifs.flat_funnel_search_batch(h.ref, qs, [45, 128], 80, 10))
            for i in range(5):
                assert bits(got[i]) == bits(unwrap(nifs.flat_funnel_search(h.ref, qs[i], [45, 128], 80, 10))), (metric, order, i)
    # a stage that overflows for one query: that query reports it, the batch call returns the first error like the loop would
    if metric == 3:
        big = GpuIndex(nifs, 3)
        xb = x[:20_000].copy()
        xb[77, :8] = 3e38
        unwrap(nifs.flat_load_matrix(big.ref, ids[:20_000], xb))
        qs = rng.uniform(-1, 1, (4, d)).astype(np.float32)
        qs[2, :8] = 3e38
        singles = [nifs.flat_funnel_search(big.ref, q, [16], 50, 5) for q in qs]
        got = nifs.flat_funnel_search_batch(big.ref, qs, [16], 50, 5)
        firsterr = next((r for r in singles if r[0] == "error"), None)
        if firsterr is not None:
            assert got == firsterr
        else:
            assert [bits(h) for h in unwrap(got)] == [bits(unwrap(r)) for r in singles]


@pytest.mark.parametrize("metric", [2, 0, 5])
def test_concurrent_funnel_callers_share_sweeps(nifs, oracle_mod, monkeypatch, metric, vt_debug):
    """funnel_search callers that meet on one handle (cosine: K6bm; L2 / manhattan: K1p) travel in groups -- only
    those with the same stages and candidates together; every answer equals the call made alone."""
    import threading
    vt_debug.set("coalesce_slots", 1)
    n, d = 40_000, 128
    x, ids = make_corpus(n, d, 4600 + metric, metric == 2, oracle_mod)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(13)
    qs = rng.uniform(-1, 1, (32, d)).astype(np.float32)
    if metric == 2:
        qs = np.stack([oracle_mod.normalize_l2(q) for q in qs])
    shapes = [([32], 100), ([16, 64], 100), ([32], 50)]
    alone = [[bits(unwrap(nifs.flat_funnel_search(g.ref, q, st, cand, 10))) for q in qs] for st, cand in shapes]
    plain = [bits(unwrap(nifs.flat_search(g.ref, q, 10))) for q in qs]
    wrong, b0 = [], nifs.flat_coalesce_stats(g.ref)

    def worker(t):
        for r in range(40):
            j = (t * 7 + r) % 32
            kind = (t + r) % 4
            if kind == 3:
                res = bits(unwrap(nifs.flat_search(g.ref, qs[j], 10))) == plain[j]
            else:
                st, cand = shapes[kind]
                res = bits(unwrap(nifs.flat_funnel_search(g.ref, qs[j], st, cand, 10))) == alone[kind][j]
            if not res:
                wrong.append((t, r, kind))
        # errors keep their own order and text
        assert nifs.flat_funnel_search(g.ref, qs[0], [d + 1], 10, 5) == ("error", "invalid prefix dimensions")

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    b1 = nifs.flat_coalesce_stats(g.ref)
    assert not wrong, wrong[:5]
    # (how many calls met is a matter of timing -- Python threads on 0.1-ms calls; callers that are MADE to meet, and the
    # sweeps they then share, are tests/test_gpu_coalesce.py::test_callers_of_three_entry_points_made_to_meet_...)
    print("coalesced: %d batches, %d calls in batches" % (b1[0] - b0[0], b1[1] - b0[1]))


def test_quantized_groups_with_massive_ties_fall_back(nifs, oracle_mod):
    """Rows drawn from two sign patterns: thousands of ties at the k-th Hamming distance, more than a
    group's per-query list holds -- the group hands over to the single-query path, same answers."""
    n, d = 30_000, 64
    rng = np.random.default_rng(516)
    base = rng.uniform(-1, 1, (2, d)).astype(np.float32)
    x = base[rng.integers(0, 2, n)] * rng.uniform(0.5, 1.5, (n, 1)).astype(np.float32)
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    g = GpuIndex(nifs, 0)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    qs = rng.uniform(-1, 1, (6, d)).astype(np.float32)
    got = unwrap(nifs.flat_quantized_search_batch(g.ref, qs, 100, 10))
    for i in range(6):
        assert bits(got[i]) == bits(unwrap(nifs.flat_quantized_search(g.ref, qs[i], 100, 10)))


def test_concurrent_quantized_callers_share_sweeps(nifs, oracle_mod, monkeypatch, vt_debug):
    """quantized_search callers that meet on one handle (collection.ex:276-295 under the read lock)
    travel in groups like plain searches do; every answer equals the call made alone."""
    import threading
    vt_debug.set("coalesce_slots", 1)
    n, d = 40_000, 128
    x, ids = make_corpus(n, d, 4200, True, oracle_mod)
    g = GpuIndex(nifs, 2)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    rng = np.random.default_rng(12)
    qs = np.stack([oracle_mod.normalize_l2(q) for q in rng.uniform(-1, 1, (32, d)).astype(np.float32)])
    alone = [bits(unwrap(nifs.flat_quantized_search(g.ref, q, 100, 10))) for q in qs]
    alone50 = [bits(unwrap(nifs.flat_quantized_search(g.ref, q, 50, 10))) for q in qs]
    plain = [bits(unwrap(nifs.flat_search(g.ref, q, 10))) for q in qs]
    wrong, b0 = [], nifs.flat_coalesce_stats(g.ref)

    def worker(t):
        for r in range(40):
            j = (t * 7 + r) % 32
            kind = (t + r) % 4
            if kind == 0:
                res = bits(unwrap(nifs.flat_search(g.ref, qs[j], 10))) == plain[j]
            elif kind == 1:
                res = bits(unwrap(nifs.flat_quantized_search(g.ref, qs[j], 50, 10))) == alone50[j]
            else:
                res = bits(unwrap(nifs.flat_quantized_search(g.ref, qs[j], 100, 10))) == alone[j]
            if not res:
                wrong.append((t, r, kind))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    b1 = nifs.flat_coalesce_stats(g.ref)
    assert not wrong, wrong[:5]
    # (how many calls met is a matter of timing -- Python threads on 0.1-ms calls -- and is measured
    # where it matters, bench.py's side.callers_quantized at N = 10 M; here only: whoever met got his own answer)
    print("coalesced: %d batches, %d calls in batches" % (b1[0] - b0[0], b1[1] - b0[1]))


@pytest.mark.parametrize("metric", [7, 8])
@pytest.mark.parametrize("d", [24, 100, 600, 768])
def test_pattern_metrics_read_the_non_zero_bits(nifs, oracle_mod, metric, d):
    """Float hamming / jaccard compare which coordinates are non-zero (distances.rs:319-347): on a
    corpus of 16 384 rows or more flat_search reads a column of non-zero bits (K4) instead of the
    rows.  Hits and scores must be the oracle's bit for bit -- single searches, limits that take
    several K4 passes, batches -- before and after inserts in any id order, upserts (zeros that
    become non-zero, -0.0 that stays zero) and deletes, which patch the column per row."""
    n = 20_000
    rng = np.random.default_rng(700 + metric + d)
    density = rng.uniform(0.05, 0.95, (n + 300, 1))
    x = (rng.uniform(-1, 1, (n + 300, d)) * (rng.uniform(0, 1, (n + 300, d)) < density)).astype(np.float32)
    x[rng.uniform(0, 1, x.shape) < 0.01] = -0.0
    x[5] = 0.0                                             # an all-zero row (jaccard: union 0 against a zero query)
    ids = [b"doc-%05d" % i for i in range(n + 300)]
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids[:n], x[:n]))
    cur = {ids[i]: x[i] for i in range(n)}
    nifs.flat_set_profiling(g.ref, True)

    def check(limits=(1, 10)):
        keys = sorted(cur)
        mat = np.stack([cur[k] for k in keys])
        packed = oracle_mod.pack_ids(keys)
        qs = [x[int(rng.integers(0, n))], np.zeros(d, np.float32),
              (rng.uniform(-1, 1, d) * (rng.uniform(0, 1, d) < 0.3)).astype(np.float32)]
        for q in qs:
            for limit in limits:
                nifs.flat_get_profile(g.ref, reset=True)
                got = unwrap(nifs.flat_search(g.ref, q, limit))
                prof = nifs.flat_get_profile(g.ref, reset=True)
                assert bits(got) == bits(oracle_mod.matrix_search(metric, mat, packed, q, limit))
                # (a boundary tie among rows that arrived out of id order may re-run once the ranks are rebuilt)
                assert prof["hamming_launches"] >= 1 and prof["scan_launches"] == 0, prof
        # batches: up to eight queries per sweep of the column (K4p) when the lists fit its wave
        # buffers (limit <= 64) and the row length has an unrolled build (not d = 600: 5 word pairs)
        batch = np.stack(qs + [x[11 + i] for i in range(16)])            # 19 queries: sweeps of 8 + 8 + 3
        for limit in (7, 64, 65):
            nifs.flat_get_profile(g.ref, reset=True)
            got = unwrap(nifs.flat_search_batch(g.ref, batch, limit))
            prof = nifs.flat_get_profile(g.ref, reset=True)
            for i in range(len(batch)):
                assert bits(got[i]) == bits(oracle_mod.matrix_search(metric, mat, packed, batch[i], limit))
            grouped = limit <= 64 and d != 600
            assert prof["hamming_queries"] == (len(batch) if grouped else 0), (limit, prof)
            assert prof["scan_launches"] == 0 or not grouped, (limit, prof)

    check(limits=(1, 10, 300, 1500))
    for step in range(4):
        for j in range(25):                                # appends in descending id order (lazy ranks)
            i = n + 299 - (step * 25 + j)
            g.insert(ids[i], x[i]); cur[ids[i]] = x[i]
        up = ids[step * 41 + 5]                            # upserts: zeros <-> non-zeros
        v = np.where(cur[up] == 0, np.float32(0.5), np.float32(-0.0)).astype(np.float32)
        g.insert(up, v); cur[up] = v
        for victim in (ids[2000 + step], ids[n - 1 - step]):
            g.delete(victim); cur.pop(victim, None)
        check()


def test_no_room_for_the_non_zero_bits_means_reading_the_rows(nifs, oracle_mod, request, monkeypatch, vt_debug):
    """The non-zero-bit column is an accelerator: when the card has no room for it the searches keep
    reading the rows, with the same hits.  (The refused allocation is injected --
    VT_TEST_REFUSE_NZBITS, libvettore_hip_hooks.so only: the test re-runs itself there.)"""
    if support.rerun_with_hooks_library(request):
        return
    vt_debug.set("test_refuse_nzbits", 1)
    n, d = 20_000, 100
    rng = np.random.default_rng(77)
    x = (rng.uniform(-1, 1, (n, d)) * (rng.uniform(0, 1, (n, d)) < 0.4)).astype(np.float32)
    ids = [b"doc-%05d" % i for i in range(n)]
    packed = oracle_mod.pack_ids(ids)
    for metric in (7, 8):
        g = GpuIndex(nifs, metric)
        unwrap(nifs.flat_load_matrix(g.ref, ids, x))
        nifs.flat_set_profiling(g.ref, True)
        for q in (x[3], x[4]):
            got = unwrap(nifs.flat_search(g.ref, q, 10))
            assert bits(got) == bits(oracle_mod.matrix_search(metric, x, packed, q, 10))
        prof = nifs.flat_get_profile(g.ref, reset=True)
        assert prof["hamming_launches"] == 0 and prof["scan_launches"] >= 2, prof


@pytest.mark.parametrize("metric", [7, 8])
def test_concurrent_pattern_metric_callers_share_sweeps(nifs, oracle_mod, metric, monkeypatch, vt_debug):
    """flat_search callers that meet on a float hamming / jaccard handle travel as a batch, and the batch
    is sweeps of the non-zero-bit column (K4p); every answer equals the call made alone and the oracle's."""
    import threading
    vt_debug.set("coalesce_slots", 1)
    n, d = 30_000, 256
    rng = np.random.default_rng(900 + metric)
    x = (rng.uniform(-1, 1, (n, d)) * (rng.uniform(0, 1, (n, d)) < 0.3)).astype(np.float32)
    ids = [b"doc-%05d" % i for i in range(n)]
    packed = oracle_mod.pack_ids(ids)
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    qs = (rng.uniform(-1, 1, (24, d)) * (rng.uniform(0, 1, (24, d)) < 0.3)).astype(np.float32)
    want = {limit: [bits(oracle_mod.matrix_search(metric, x, packed, q, limit)) for q in qs] for limit in (10, 100)}
    for limit in (10, 100):
        assert [bits(unwrap(nifs.flat_search(g.ref, q, limit))) for q in qs] == want[limit]
    wrong = []

    def worker(t):
        for r in range(30):
            j = (t * 5 + r) % 24
            limit = 100 if (t + r) % 5 == 0 else 10          # (lists of 100 do not fit K4p: those callers go alone)
            if bits(unwrap(nifs.flat_search(g.ref, qs[j], limit))) != want[limit][j]:
                wrong.append((t, r, limit))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not wrong, wrong[:5]


def test_pattern_metric_batches_beyond_one_group_call(nifs, oracle_mod):
    """300 queries under float hamming: K4p sweeps in two calls of at most 256 queries (the last one a
    partial sweep), every list the oracle's."""
    n, d = 17_000, 64
    rng = np.random.default_rng(4242)
    x = (rng.uniform(-1, 1, (n, d)) * (rng.uniform(0, 1, (n, d)) < 0.4)).astype(np.float32)
    ids = [b"doc-%05d" % i for i in range(n)]
    packed = oracle_mod.pack_ids(ids)
    g = GpuIndex(nifs, 7)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    qs = (rng.uniform(-1, 1, (300, d)) * (rng.uniform(0, 1, (300, d)) < 0.4)).astype(np.float32)
    nifs.flat_set_profiling(g.ref, True)
    got = unwrap(nifs.flat_search_batch(g.ref, qs, 5))
    prof = nifs.flat_get_profile(g.ref, reset=True)
    assert prof["hamming_queries"] == 300 and prof["hamming_launches"] == 32 + 6 and prof["scan_launches"] == 0, prof
    for i in range(300):
        assert bits(got[i]) == bits(oracle_mod.matrix_search(7, x, packed, qs[i], 5))


@pytest.mark.parametrize("metric", [7, 8])
def test_funnel_under_float_hamming_and_jaccard_reads_the_non_zero_bits(nifs, oracle_mod, metric):
    """funnel_search on a float hamming / jaccard collection (collection.ex:245-260 -> search.rs:38-73 on prefixes):
    the stage over ALL rows looks at nothing but which of the first `stage` coordinates are non-zero
    (distances.rs:319-347), so it reads that prefix of the non-zero-bit column (K4 with a prefix mask) instead of
    the rows; later stages and the rerank gather rows.  Prefixes on and off word and word-pair borders; equal to
    the oracle's composition bit for bit, and to the same search with the column switched off."""
    n, d = 30_000, 200
    rng = np.random.default_rng(60 + metric)
    x = (rng.uniform(-1, 1, (n, d)) * (rng.uniform(0, 1, (n, d)) < 0.45)).astype(np.float32)
    x[500:530] = x[500]
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    g = GpuIndex(nifs, metric)
    unwrap(nifs.flat_load_matrix(g.ref, ids, x))
    nifs.flat_set_profiling(g.ref, True)
    rows = [(ids[i], x[i]) for i in range(n)]
    by_id = dict(rows)
    for stages, cand, limit in (([64], 100, 10), ([100], 60, 5), ([129, 200], 256, 20), ([1], 30, 30), ([200], 10, 10), ([13, 70], 300, 7)):
        q = (rng.uniform(-1, 1, d) * (rng.uniform(0, 1, d) < 0.45)).astype(np.float32)
        if stages == [64]:
            q = x[500]
        cur = rows
        for st in stages:
            kept = oracle_mod.vector_top_k(cur, q, metric, st, cand)
            cur = [(i, by_id[i]) for i, _ in kept]
        want = oracle_mod.vector_top_k(cur, q, metric, d, limit)
        got = unwrap(nifs.flat_funnel_search(g.ref, q, stages, cand, limit))
        assert bits(got) == bits(want), (metric, stages, cand, limit)
    prof = nifs.flat_get_profile(g.ref)
    # every first stage over a true prefix was a pass over the bit column (K4, or its host-staged form for 300 candidates);
    # [200] is the whole row and goes the collection's ordinary way
    assert prof["hamming_launches"] + prof["prefix_launches"] >= 5, prof
    # ... and what those passes read is the bit words (8 B per 64 coordinates and row), not the rows' prefixes
    # (4 B per coordinate and row: 36.8 MB over these six searches)
    assert prof["prefix_bytes"] + prof["hamming_bytes"] < n * 64 * 4, prof
