"""Pins the CPU oracle against every known-answer test the reference holds for
the flat-index hot path (fixtures in tests/golden/).

Each test runs under all four candidate `wide::f32x8::reduce_add` lane orders:
the reference's known answers do not depend on that order, which is exactly
why the order itself stays unpinned (see oracle/vt_oracle.h).
"""
import math

import numpy as np
import pytest

import support
from support import b, close, load, run_steps, same_f32, full_sort

ORDERS = [0, 1, 2, 3]
F32_MAX = float(np.finfo(np.float32).max)


@pytest.fixture(params=ORDERS, ids=["pair", "avx", "seq", "sse2"])
def orc(request, oracle_mod):
    oracle_mod.set_reduce_order(request.param)
    yield oracle_mod
    oracle_mod.set_reduce_order(oracle_mod.DEFAULT_ORDER)


def code(orc, name):
    return orc.METRIC_CODE[name]


# ------------------------------------------------------------------ flat.rs
def test_flat_rs_scripts(orc):
    for case in load("flat_rs.json"):
        if case.get("differential"):
            continue
        ix = orc.FlatIndex(code(orc, case["metric"]))
        run_steps(ix, case["steps"], orc.OracleError)


def test_flat_rs_heap_matches_full_sort_all_metrics(orc):
    case = next(c for c in load("flat_rs.json") if c.get("differential"))
    rows, q = case["rows"], case["query"]
    for name in case["metrics"]:
        m = code(orc, name)
        ix = orc.FlatIndex(m)
        ix.insert_many([(r[0], r[1]) for r in rows])
        for limit in case["limits"]:
            expected = full_sort(rows, lambda v: orc.compute(m, q, v), lambda raw: orc.rank_value(m, raw), limit)
            got = ix.search(q, limit)
            assert [(g[0], np.float32(g[1]).tobytes()) for g in got] == \
                   [(e[0], np.float32(e[1]).tobytes()) for e in expected], (name, limit)


# ------------------------------------------------------------- distances.rs
def test_distances_metric_values(orc):
    d = load("distances_rs.json")
    c = d["computes_every_metric_and_rank_semantics"]
    for name, want in c["exact"].items():
        assert orc.compute(code(orc, name), c["left"], c["right"]) == want, name
    for name, (want, tol) in c["close"].items():
        assert abs(orc.compute(code(orc, name), c["left"], c["right"]) - want) < tol, name
    for name, raw, want in c["rank_value"]:
        assert orc.rank_value(code(orc, name), raw) == want


def test_distances_validation_and_normalize(orc):
    c = load("distances_rs.json")["validates_dimensions_normalization_and_finite_values"]
    for name, l, r, msg in c["compute_errors"]:
        with pytest.raises(orc.OracleError, match=msg):
            orc.compute(code(orc, name), l, r)
    for v, want in c["normalize_l2"]:
        assert list(orc.normalize_l2(v)) == [np.float32(x) for x in want]
    for l, r, want in c["cosine"]:
        assert orc.cosine(l, r) == want
    for v, want, tol in c["normalize_l2_close"]:
        got = orc.normalize_l2(v)
        assert all(abs(float(g) - w) < tol for g, w in zip(got, want))
    for name, l, r, msg in c["compute_checked_errors"]:
        with pytest.raises(orc.OracleError, match=msg):
            orc.compute(code(orc, name), l, r, checked=True)


def test_distances_sign_bits_and_packed(orc):
    c = load("distances_rs.json")["packs_bits_and_masks_unused_coordinates"]
    for v, words in c["compress"]:
        assert list(orc.compress_sign_bits(v)) == words
    for l, r, dims, want in c["packed_hamming"]:
        assert orc.packed_hamming(l, r, dims) == want
    for l, r, dims, want in c["packed_jaccard"]:
        assert orc.packed_jaccard(l, r, dims) == want
    for l, r, dims, msg in c["packed_errors"]:
        with pytest.raises(orc.OracleError, match=msg):
            orc.packed_hamming(l, r, dims)


def test_distances_metric_codes(orc):
    c = load("distances_rs.json")["decodes_metric_codes"]
    for k, name in c["codes"].items():
        assert orc.METRICS[int(k)] == name
        assert orc.FlatIndex(int(k)).metric == int(k)
    for bad in c["invalid"]:
        with pytest.raises(orc.OracleError, match="unknown metric"):
            orc.vector_top_k([], [1.0], bad, 1, 1)


def test_distances_simd_and_tail_match_f64(orc):
    c = load("distances_rs.json")["simd_and_tail_kernels_match_scalar_oracles"]
    tol = c["tolerance"]
    for v in c["vectors"]:
        l = np.asarray(v["left"], dtype=np.float32)
        r = np.asarray(v["right"], dtype=np.float32)
        l64, r64 = l.astype(np.float64), r.astype(np.float64)
        # the reference test sums sequentially in f64, then narrows to f32
        exp_dot = float(np.float32(sum((x * y for x, y in zip(l64, r64)), 0.0)))
        exp_l2 = float(np.float32(sum(((x - y) ** 2 for x, y in zip(l64, r64)), 0.0)))
        exp_man = float(np.float32(sum((abs(x - y) for x, y in zip(l64, r64)), 0.0)))
        exp_cheb = float(np.max(np.abs(l - r))) if len(l) else 0.0
        n = v["len"]
        assert close(float(orc.compute(3, l, r)), exp_dot, tol), n
        assert close(float(orc.compute(1, l, r)), exp_l2, tol), n
        assert close(float(orc.compute(5, l, r)), exp_man, tol), n
        assert float(orc.compute(6, l, r)) == exp_cheb, n


def test_distances_overflow_recovery(orc):
    c = load("distances_rs.json")["recovers_representable_results_after_f32_intermediate_overflow"]
    for name, l, r, want, tol in c["close"]:
        assert close(float(orc.compute(code(orc, name), l, r)), want, tol)
    for name, l, r, want, sign in c["exact"]:
        got = orc.compute(code(orc, name), l, r)
        assert same_f32(got, want), (name, got)
        assert math.copysign(1.0, float(got)) == (1.0 if sign == "+" else -1.0)
    for name, l, r in c["errors"]:
        with pytest.raises(orc.OracleError, match="metric overflow"):
            orc.compute(code(orc, name), l, r)


def test_distances_cosine_and_normalize_invariants(orc):
    c = load("distances_rs.json")["cosine_and_normalization_obey_numerical_invariants"]
    for l, r, want in c["cosine_exact"]:
        assert orc.cosine(l, r) == want
    for l, r, msg in c["cosine_errors"]:
        with pytest.raises(orc.OracleError, match=msg):
            orc.cosine(l, r)
    for l, r, want, tol in c["cosine_close"]:
        assert close(float(orc.cosine(l, r)), want, tol)
    assert list(orc.normalize_l2([])) == []
    for v in c["normalize_l2_unit"]:
        n = orc.normalize_l2(v).astype(np.float64)
        assert close(float(np.float32(np.sum(n * n))), 1.0, 1e-6)
    for bad in c["non_finite"]:
        with pytest.raises(orc.OracleError, match="non-finite"):
            orc.normalize_l2([bad])
        with pytest.raises(orc.OracleError, match="non-finite"):
            orc.cosine([bad], [1.0])


def test_distances_packed_word_boundaries(orc):
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
            right[words - 1] ^= (~used) & full
        assert orc.packed_hamming(left, right, dims) == float(len(flipped))
        assert close(float(orc.packed_jaccard(left, right, dims)), len(flipped) / dims, 1e-6)
    l, r, dims, want = c["jaccard_zero"]
    assert orc.packed_jaccard(l, r, dims) == want
    l, r, dims = c["jaccard_error"]
    with pytest.raises(orc.OracleError):
        orc.packed_jaccard(l, r, dims)


# ---------------------------------------------------------------- search.rs
def _vec_call(orc, vectors, call):
    return orc.vector_top_k([(v[0], v[1]) for v in vectors], call["query"], code(orc, call["metric"]),
                            call["dimensions"], call["limit"])


def _check_call(orc, fn, call):
    if "expect_error" in call:
        with pytest.raises(orc.OracleError, match=call["expect_error"]):
            fn()
        return
    hits = fn()
    if "expect" in call:
        assert hits == [(b(e[0]), e[1]) for e in call["expect"]]
    if "expect_first_id" in call:
        assert hits[0][0] == b(call["expect_first_id"])


def test_search_rs_vector_top_k_cases(orc):
    d = load("search_rs.json")
    c = d["vector_top_k_handles_prefixes_similarity_and_ties"]
    for call in c["calls"]:
        _check_call(orc, lambda: _vec_call(orc, c["vectors"], call), call)
    for key in ("vector_top_k_rejects_bad_dimensions_and_values",
                "vector_top_k_validates_queries_and_only_reads_the_requested_prefix"):
        for call in d[key]["calls"]:
            _check_call(orc, lambda: _vec_call(orc, call["vectors"], call), call)
    c = d["stable_ties_do_not_depend_on_candidate_order"]
    for vectors in (c["forward"], list(reversed(c["forward"]))):
        _check_call(orc, lambda: _vec_call(orc, vectors, c), c)


def test_search_rs_vector_top_k_matches_full_sort(orc):
    c = load("search_rs.json")["vector_top_k_matches_full_sort_for_every_metric_and_limit"]
    rows, q = c["rows"], c["query"]
    for name in c["metrics"]:
        m = code(orc, name)
        for dims in c["dimensions"]:
            def raw_fn(v):
                if m == 2:
                    return orc.cosine(q[:dims], v[:dims])
                return orc.compute(m, q[:dims], v[:dims])
            for limit in c["limits"]:
                expected = full_sort(rows, raw_fn, lambda raw: orc.rank_value(m, raw), limit)
                got = orc.vector_top_k([(r[0], r[1]) for r in rows], q, m, dims, limit)
                assert [(g[0], np.float32(g[1]).tobytes()) for g in got] == \
                       [(e[0], np.float32(e[1]).tobytes()) for e in expected], (name, dims, limit)


def test_search_rs_binary_top_k(orc):
    d = load("search_rs.json")
    c = d["binary_top_k_masks_padding_and_orders_ids"]
    q = orc.compress_sign_bits(c["query_vector"])
    vecs = [(v[0], orc.compress_sign_bits(v[1])) for v in c["vectors"]]
    assert orc.binary_top_k(vecs, q, c["dimensions"], c["limit"]) == [(b(e[0]), e[1]) for e in c["expect"]]
    for call in d["binary_top_k_validates_empty_batches_limits_and_word_boundaries"]["calls"]:
        _check_call(orc, lambda: orc.binary_top_k([(v[0], v[1]) for v in call["vectors"]], call["query"],
                                                  call["dimensions"], call["limit"]), call)


# ------------------------------------------------- Elixir tests through the NIF
def _prep(orc, metric_name, v):
    """Collection.prepare_embedding / prepare_query: cosine collections
    L2-normalize (collection.ex:1317-1319, :352-357)."""
    return orc.normalize_l2(v) if metric_name == "cosine" else np.asarray(v, dtype=np.float32)


def test_elixir_all_metrics_stable_top_k(orc):
    c = load("elixir_nif.json")["all_supported_metrics_return_stable_top_k_results"]
    for name in c["metrics"]:
        ix = orc.FlatIndex(code(orc, name))
        ix.insert_many([(r[0], _prep(orc, name, r[1])) for r in c["rows"]])
        hits = ix.search(_prep(orc, name, c["query"]), c["limit"])
        assert [h[0] for h in hits] == [b(x) for x in c["expect_ids"]], name


def test_elixir_batched_helpers(orc):
    c = load("elixir_nif.json")["batched_native_helpers"]
    vecs = [(v[0], v[1]) for v in c["vectors"]]
    for mc in c["metric_codes"]:
        hits = orc.vector_top_k(vecs, c["query"], mc, c["dimensions"], c["limit"])
        assert [h[0] for h in hits] == [b(x) for x in c["expect_ids"]], mc
    with pytest.raises(orc.OracleError, match=c["unknown_metric"][1]):
        orc.vector_top_k(vecs, c["query"], c["unknown_metric"][0], 2, 2)
    with pytest.raises(orc.OracleError, match=c["bad_prefix"][2]):
        orc.vector_top_k(vecs, c["query"], c["bad_prefix"][0], c["bad_prefix"][1], 2)
    bn = c["binary"]
    assert orc.binary_top_k([(v[0], v[1]) for v in bn["vectors"]], bn["query"], bn["dimensions"], bn["limit"]) == \
        [(b(e[0]), e[1]) for e in bn["expect"]]


def test_elixir_cosine_collection_first_hit(orc):
    c = load("elixir_nif.json")["cosine_collection_result_semantics"]
    ix = orc.FlatIndex(code(orc, "cosine"))
    ix.insert_many([(r[0], orc.normalize_l2(r[1])) for r in c["rows"]])
    hits = ix.search(orc.normalize_l2(c["query"]), c["limit"])
    assert hits[0][0] == b(c["expect_first"]["id"])
    raw = hits[0][1]
    assert raw == c["expect_first"]["score"] and 1.0 - raw == c["expect_first"]["distance"]


def _quantized(orc, metric_name, rows, query, candidates, limit):
    """collection.ex:276-295: sign-bit candidates (binary_top_k) then exact rerank (vector_top_k)."""
    m = code(orc, metric_name)
    prepared = [(r[0], _prep(orc, metric_name, r[1])) for r in rows]
    q = _prep(orc, metric_name, query)
    dims = len(q)
    cands = orc.binary_top_k([(i, orc.compress_sign_bits(v)) for i, v in prepared], orc.compress_sign_bits(q),
                             dims, candidates)
    by_id = {b(i): v for i, v in prepared}
    return orc.vector_top_k([(cid, by_id[cid]) for cid, _ in cands], q, m, dims, limit)


def test_elixir_binary_quantized_search(orc):
    c = load("elixir_nif.json")["binary_quantized_search"]
    name, words = c["binary_vector_of"]
    vec = dict((r[0], r[1]) for r in c["rows"])[name]
    assert list(orc.compress_sign_bits(vec)) == words
    hits = _quantized(orc, c["metric"], c["rows"], c["query"], c["candidates"], c["limit"])
    assert [(h[0], h[1]) for h in hits] == [(b(e["id"]), e["distance"]) for e in c["expect"]]


def test_elixir_full_candidate_quantized_equals_flat(orc):
    c = load("elixir_nif.json")["full_candidate_adaptive_modes_agree_with_exact_flat_search"]
    ix = orc.FlatIndex(code(orc, c["metric"]))
    ix.insert_many([(r[0], r[1]) for r in c["rows"]])
    exact = ix.search(c["query"], c["limit"])
    quant = _quantized(orc, c["metric"], c["rows"], c["query"], c["candidates"], c["limit"])
    assert [h[0] for h in quant] == [h[0] for h in exact]


# ----------------------------------------------- summation-order bookkeeping
def test_reduce_orders_differ_only_in_last_bits(oracle_mod):
    rng = np.random.default_rng(7)
    a = rng.uniform(-1, 1, 768).astype(np.float32)
    c = rng.uniform(-1, 1, 768).astype(np.float32)
    vals = []
    for o in ORDERS:
        oracle_mod.set_reduce_order(o)
        vals.append(float(oracle_mod.compute(3, a, c)))
    oracle_mod.set_reduce_order(oracle_mod.DEFAULT_ORDER)
    exact = float(np.dot(a.astype(np.float64), c.astype(np.float64)))
    assert all(close(v, exact, 2e-6) for v in vals)


def test_order_probe_separates_the_four_orders(oracle_mod):
    """the probe a maintainer runs on the reference build."""
    import order_probe
    t = order_probe.table()
    assert {k: v[0] for k, v in t.items()} == {"PAIR": 0xbfefdf3c, "AVX": 0x40480000, "SEQ": 0x3f9020c4,
                                                "SSE2": 0x40081064}
