#!/usr/bin/env python3
"""Writes tests/golden/*.json: the known-answer tests the reference holds for the
flat-index hot path, transcribed as DATA (inputs + expected outputs).

Nothing here is computed by the oracle or the product: expected values are the
literals the reference's own tests assert (file:line cited per case).  Where a
reference test builds its inputs from a formula, the formula is re-evaluated
here in the same arithmetic (f32 for Rust `as f32` expressions, f64 then
narrowed to f32 for Elixir floats that pass through the NIF decoder).

Run:  python tests/golden/make_fixtures.py      (idempotent)
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
F32_MAX = float(np.finfo(np.float32).max)
USIZE_MAX = (1 << 64) - 1
f32 = np.float32

METRICS = ["l2", "l2_squared", "cosine", "inner_product", "negative_inner_product",
           "manhattan", "chebyshev", "hamming", "jaccard"]


def fl(x):
    """f32 value as an exactly representable python float."""
    return float(f32(x))


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, allow_nan=True)
        f.write("\n")


# --------------------------------------------------------------------------
# flat.rs unit tests (native/vettore/src/flat.rs:146-304)
# --------------------------------------------------------------------------
def flat_rs():
    cases = []
    cases.append({
        "name": "inserts_replaces_deletes_and_returns_stable_top_k",
        "cite": "flat.rs:164-180",
        "metric": "l2",
        "steps": [
            {"op": "insert", "id": "b", "vector": [2.0]},
            {"op": "insert", "id": "a", "vector": [0.0]},
            {"op": "insert", "id": "c", "vector": [2.0]},
            {"op": "search", "query": [1.0], "limit": 2, "expect": [["a", 1.0], ["b", 1.0]]},
            {"op": "insert", "id": "a", "vector": [10.0]},
            {"op": "search", "query": [2.0], "limit": 1, "expect_first_id": "b"},
            {"op": "delete", "id": "b"},
            {"op": "search", "query": [2.0], "limit": 1, "expect_first_id": "c"},
        ],
    })
    cases.append({
        "name": "batch_validation_is_atomic",
        "cite": "flat.rs:182-196",
        "metric": "inner_product",
        "steps": [
            {"op": "insert", "id": "existing", "vector": [1.0, 0.0]},
            {"op": "insert_many", "items": [["valid", [0.0, 1.0]], ["invalid", [1.0]]],
             "expect_error": "dimension mismatch"},
            {"op": "expect_len", "len": 1},
            {"op": "search", "query": [0.0, 1.0], "limit": 10, "expect_ids": ["existing"]},
            {"op": "insert", "id": "nan", "vector": [float("nan"), 0.0],
             "expect_error": "vector contains a non-finite value"},
        ],
    })
    cases.append({
        "name": "rejects_invalid_queries_and_handles_empty_limits",
        "cite": "flat.rs:198-206",
        "metric": "cosine",
        "steps": [
            {"op": "insert", "id": "empty", "vector": [], "expect_error": "vector must not be empty"},
            {"op": "insert", "id": "a", "vector": [1.0, 0.0]},
            {"op": "search", "query": [1.0], "limit": 1, "expect_error": "dimension mismatch"},
            {"op": "search", "query": [float("inf"), 0.0], "limit": 1,
             "expect_error": "vector contains a non-finite value"},
            {"op": "search", "query": [1.0, 0.0], "limit": 0, "expect": []},
        ],
    })
    cases.append({
        "name": "empty_batches_unknown_deletes_and_dimension_resets_are_total",
        "cite": "flat.rs:251-267",
        "metric": "l2",
        "steps": [
            {"op": "insert_many", "items": []},
            {"op": "search", "query": [1.0], "limit": 10, "expect": []},
            {"op": "delete", "id": "missing"},
            {"op": "insert", "id": "one", "vector": [1.0]},
            {"op": "delete", "id": "missing"},
            {"op": "expect_dimension", "dimension": 1},
            {"op": "delete", "id": "one"},
            {"op": "expect_dimension", "dimension": None},
            {"op": "insert", "id": "two", "vector": [1.0, 2.0]},
            {"op": "expect_dimension", "dimension": 2},
            {"op": "search", "query": [1.0, 2.0], "limit": USIZE_MAX, "expect_len": 1},
        ],
    })
    cases.append({
        "name": "duplicate_batch_ids_replace_deterministically_and_large_l2_stays_finite",
        "cite": "flat.rs:269-281",
        "metric": "l2",
        "steps": [
            {"op": "insert_many", "items": [["same", [0.0]], ["same", [fl(1.0e20)]]]},
            {"op": "expect_len", "len": 1},
            {"op": "search", "query": [0.0], "limit": 1, "expect_first_id": "same",
             "expect_finite": True, "expect_close": [["same", fl(1.0e20)]], "rel_tol": 1e-6},
        ],
    })
    # exact_heap_matches_a_full_sort_for_all_metrics (flat.rs:208-249): the
    # test is differential (heap == full sort by (rank.total_cmp, id) of
    # compute()); the fixture is its generated input.
    rows = []
    for index in range(51):
        v = [
            (f32(index) - f32(25.0)) / f32(9.0),
            (f32(index * 13 % 31) - f32(15.0)) / f32(7.0),
            f32(0.0) if index % 2 == 0 else f32(1.0),
        ]
        rows.append(["v-%02d" % index, [float(x) for x in v]])
    cases.append({
        "name": "exact_heap_matches_a_full_sort_for_all_metrics",
        "cite": "flat.rs:208-249",
        "differential": True,
        "metrics": METRICS,
        "rows": rows,
        "query": [0.5, -1.25, 1.0],
        "limits": [1, 7, 51, 100],
    })
    # FlatHit ordering (flat.rs:283-303): ("a",rank 1.0) == ("a",rank 1.0, other raw); ("a",1) < ("b",1)
    cases.append({
        "name": "heap_hit_equality_and_partial_order_include_the_external_id",
        "cite": "flat.rs:283-303",
        "metric": "l2",
        "steps": [
            {"op": "insert", "id": "b", "vector": [1.0]},
            {"op": "insert", "id": "a", "vector": [1.0]},
            {"op": "search", "query": [0.0], "limit": 2, "expect": [["a", 1.0], ["b", 1.0]]},
        ],
    })
    dump("flat_rs.json", cases)


# --------------------------------------------------------------------------
# distances.rs unit tests (native/vettore/src/distances.rs:483-708)
# --------------------------------------------------------------------------
def distances_rs():
    out = {}
    out["computes_every_metric_and_rank_semantics"] = {
        "cite": "distances.rs:495-515",
        "left": [1.0, 0.0, 1.0], "right": [0.0, 1.0, 1.0],
        "exact": {"l2_squared": 2.0, "cosine": 1.0, "inner_product": 1.0,
                  "negative_inner_product": -1.0, "manhattan": 2.0, "chebyshev": 1.0,
                  "hamming": 2.0},
        "close": {"l2": [float(np.sqrt(f32(2.0))), 1e-6], "jaccard": [2.0 / 3.0, 1e-6]},
        "rank_value": [["inner_product", 2.0, -2.0], ["cosine", 0.25, 0.75]],
    }
    out["validates_dimensions_normalization_and_finite_values"] = {
        "cite": "distances.rs:517-537",
        "compute_errors": [["l2", [1.0], [1.0, 2.0], "dimension mismatch"]],
        "normalize_l2": [[[3.0, 4.0], [fl(0.6), fl(0.8)]], [[0.0, 0.0], [0.0, 0.0]]],
        "cosine": [[[2.0, 0.0], [4.0, 0.0], 1.0], [[0.0, 0.0], [4.0, 0.0], 0.0]],
        "normalize_l2_close": [[[F32_MAX, F32_MAX], [float(np.sqrt(0.5)), float(np.sqrt(0.5))], 1e-6]],
        "compute_checked_errors": [
            ["inner_product", [F32_MAX], [F32_MAX], "metric overflow"],
            ["hamming", [float("nan")], [0.0], "vector contains a non-finite value"],
        ],
    }
    out["packs_bits_and_masks_unused_coordinates"] = {
        "cite": "distances.rs:539-548",
        "compress": [[[1.0, -1.0, 0.0], [5]], [[-1.0, -1.0, 0.0], [4]]],
        "packed_hamming": [[[5], [4], 3, 1.0]],
        "packed_jaccard": [[[5], [4], 3, 0.5]],
        "packed_errors": [[[5], [4], 0, "dimensions must be positive"],
                          [[5], [], 3, "dimension mismatch"]],
    }
    out["decodes_metric_codes"] = {
        "cite": "distances.rs:550-568",
        "codes": {str(i): m for i, m in enumerate(METRICS)},
        "invalid": [9, 255],
    }
    # simd_and_tail_kernels_match_scalar_oracles (distances.rs:570-609):
    # generated f32 inputs for len 0..40; expectation = f64 reference at 2e-6
    # relative (scale max(1,|a|,|b|)); chebyshev exact.
    tails = []
    for n in range(41):
        left = [float((f32(i * 37 % 23) - f32(11.0)) / f32(3.0)) for i in range(n)]
        right = [float((f32(i * 19 % 29) - f32(14.0)) / f32(5.0)) for i in range(n)]
        tails.append({"len": n, "left": left, "right": right})
    out["simd_and_tail_kernels_match_scalar_oracles"] = {
        "cite": "distances.rs:570-609", "tolerance": 2.0e-6, "vectors": tails}
    out["recovers_representable_results_after_f32_intermediate_overflow"] = {
        "cite": "distances.rs:611-635",
        "close": [["l2", [fl(1.0e20)], [0.0], fl(1.0e20), 1e-6]],
        "exact": [
            ["inner_product", [F32_MAX, F32_MAX], [2.0, -2.0], 0.0, "+"],
            ["negative_inner_product", [F32_MAX, F32_MAX], [2.0, -2.0], -0.0, "-"],
            ["jaccard", [0.0, 0.0], [0.0, 0.0], 0.0, "+"],
        ],
        "errors": [
            ["l2_squared", [fl(1.0e20)], [0.0]],
            ["l2", [F32_MAX, F32_MAX], [0.0, 0.0]],
            ["manhattan", [F32_MAX, F32_MAX], [0.0, 0.0]],
            ["chebyshev", [F32_MAX], [-F32_MAX]],
        ],
    }
    out["cosine_and_normalization_obey_numerical_invariants"] = {
        "cite": "distances.rs:637-673",
        "cosine_exact": [[[], [], 0.0]],
        "cosine_errors": [[[1.0], [1.0, 2.0], "dimension mismatch"]],
        "cosine_close": [[[2.0, 0.0], [-5.0, 0.0], -1.0, 1e-6], [[3.0, 4.0], [6.0, 8.0], 1.0, 1e-6]],
        "normalize_l2_empty": [],
        "normalize_l2_unit": [[3.0, -4.0, 12.0]],
        "non_finite": [float("nan"), float("inf"), float("-inf")],
    }
    out["packed_distances_cover_word_boundaries_and_ignore_padding"] = {
        "cite": "distances.rs:675-707",
        "dimensions": [1, 63, 64, 65, 127, 128, 129],
        "jaccard_zero": [[0], [0], 64, 0.0],
        "jaccard_error": [[], [], 1],
    }
    dump("distances_rs.json", out)


# --------------------------------------------------------------------------
# search.rs unit tests (native/vettore/src/search.rs:112-304)
# --------------------------------------------------------------------------
def search_rs():
    out = {}
    out["vector_top_k_handles_prefixes_similarity_and_ties"] = {
        "cite": "search.rs:158-174",
        "vectors": [["b", [1.0, 10.0]], ["a", [1.0, -10.0]], ["c", [-1.0, 0.0]]],
        "calls": [
            {"query": [1.0, 0.0], "metric": "l2", "dimensions": 1, "limit": 2,
             "expect": [["a", 0.0], ["b", 0.0]]},
            {"query": [1.0, 1.0], "metric": "inner_product", "dimensions": 2, "limit": 1,
             "expect_first_id": "b"},
        ],
    }
    out["vector_top_k_rejects_bad_dimensions_and_values"] = {
        "cite": "search.rs:176-184",
        "calls": [
            {"vectors": [], "query": [1.0], "metric": "l2", "dimensions": 0, "limit": 1,
             "expect_error": "invalid prefix dimensions"},
            {"vectors": [["a", [1.0]]], "query": [1.0, 2.0], "metric": "l2", "dimensions": 2, "limit": 1,
             "expect_error": "dimension mismatch"},
            {"vectors": [["a", [float("nan")]]], "query": [1.0], "metric": "l2", "dimensions": 1, "limit": 1,
             "expect_error": "vector contains a non-finite value"},
        ],
    }
    out["binary_top_k_masks_padding_and_orders_ids"] = {
        "cite": "search.rs:186-203",
        "query_vector": [1.0, -1.0, 1.0],
        "vectors": [["b", [1.0, 1.0, 1.0]], ["a", [1.0, -1.0, 1.0]]],
        "dimensions": 3, "limit": 2,
        "expect": [["a", 0.0], ["b", 1.0]],
    }
    rows = []
    for index in range(37):
        v = [
            (f32(index) - f32(18.0)) / f32(7.0),
            (f32(index * 11 % 17) - f32(8.0)) / f32(5.0),
            (f32(index * 7 % 13) - f32(6.0)) / f32(3.0),
            f32(0.0) if index % 3 == 0 else f32(1.0),
        ]
        rows.append(["id-%02d" % index, [float(x) for x in v]])
    out["vector_top_k_matches_full_sort_for_every_metric_and_limit"] = {
        "cite": "search.rs:205-232", "differential": True,
        "rows": rows, "query": [0.25, -0.75, 1.5, 0.0],
        "metrics": METRICS, "dimensions": [1, 3, 4], "limits": [0, 1, 5, 37, 100],
    }
    out["vector_top_k_validates_queries_and_only_reads_the_requested_prefix"] = {
        "cite": "search.rs:234-244",
        "calls": [
            {"vectors": [], "query": [float("nan")], "metric": "l2", "dimensions": 1, "limit": 1,
             "expect_error": "vector contains a non-finite value"},
            {"vectors": [], "query": [1.0], "metric": "l2", "dimensions": 2, "limit": 1,
             "expect_error": "invalid prefix dimensions"},
            {"vectors": [["a", [1.0, float("nan")]]], "query": [1.0, float("nan")], "metric": "l2",
             "dimensions": 1, "limit": 1, "expect": [["a", 0.0]]},
        ],
    }
    out["binary_top_k_validates_empty_batches_limits_and_word_boundaries"] = {
        "cite": "search.rs:246-260",
        "calls": [
            {"vectors": [], "query": [], "dimensions": 0, "limit": 1, "expect_error": "dimensions must be positive"},
            {"vectors": [], "query": [], "dimensions": 1, "limit": 1, "expect_error": "dimension mismatch"},
            {"vectors": [], "query": [0], "dimensions": 1, "limit": 1, "expect": []},
            {"vectors": [["same", [USIZE_MAX, 1]], ["far", [0, 0]]], "query": [USIZE_MAX, 1],
             "dimensions": 65, "limit": 0, "expect": []},
            {"vectors": [["same", [USIZE_MAX, 1]], ["far", [0, 0]]], "query": [USIZE_MAX, 1],
             "dimensions": 65, "limit": 10, "expect": [["same", 0.0], ["far", 65.0]]},
            {"vectors": [["bad", [0]]], "query": [USIZE_MAX, 1], "dimensions": 65, "limit": 1,
             "expect_error": "dimension mismatch"},
        ],
    }
    out["stable_ties_do_not_depend_on_candidate_order"] = {
        "cite": "search.rs:262-281",
        "forward": [["c", [1.0]], ["a", [1.0]], ["b", [1.0]]],
        "query": [1.0], "metric": "l2", "dimensions": 1, "limit": 2,
        "expect": [["a", 0.0], ["b", 0.0]],
    }
    dump("search_rs.json", out)


# --------------------------------------------------------------------------
# Elixir tests that go through the real NIF (test/*.exs)
# --------------------------------------------------------------------------
def elixir_nif():
    out = {}
    out["all_supported_metrics_return_stable_top_k_results"] = {
        "cite": "test/vector_algorithms_hardening_test.exs:20-36",
        "metrics": METRICS,
        "note": "Collection default normalize is :l2 for :cosine, :none otherwise (collection.ex:1317-1319)",
        "rows": [["b", [0.0, 1.0]], ["a", [1.0, 0.0]], ["c", [1.0, 0.0]]],
        "query": [1.0, 0.0], "limit": 2, "expect_ids": ["a", "c"],
    }
    out["phantom_native_id_and_ok_unit"] = {
        "cite": "test/vector_algorithms_hardening_test.exs:53-57",
        "metric": "l2",
        "flat_insert": ["phantom", [0.0]], "expect_insert": ["ok", []],
        "put_empty_error": "vector must not be empty",
    }
    out["batched_native_helpers"] = {
        "cite": "test/vector_algorithms_hardening_test.exs:90-106",
        "vectors": [["b", [1.0, 0.0]], ["a", [1.0, 0.0]], ["c", [0.0, 1.0]]],
        "query": [1.0, 0.0], "metric_codes": list(range(9)), "dimensions": 2, "limit": 2,
        "expect_ids": ["a", "b"],
        "unknown_metric": [9, "unknown metric"],
        "bad_prefix": [0, 0, "invalid prefix dimensions"],
        "binary": {"vectors": [["b", [1]], ["a", [3]]], "query": [3], "dimensions": 2, "limit": 2,
                   "expect": [["a", 0.0], ["b", 1.0]]},
    }
    out["cosine_collection_result_semantics"] = {
        "cite": "test/vector_db_test.exs:26-53",
        "metric": "cosine", "normalize": "l2", "score": "raw",
        "rows": [["right", [1.0, 0.0]], ["up", [0.0, 1.0]], ["left", [-1.0, 0.0]]],
        "query": [1.0, 0.0], "limit": 2,
        "expect_first": {"id": "right", "score": 1.0, "distance": 0.0},
    }
    out["binary_quantized_search"] = {
        "cite": "test/vector_db_test.exs:154-174",
        "metric": "l2",
        "rows": [["exact", [1.0, 1.0]], ["same_bits_far", [100.0, 100.0]], ["opposite", [-1.0, -1.0]]],
        "binary_vector_of": ["exact", [3]],
        "query": [1.0, 1.0], "candidates": 2, "limit": 1,
        "expect": [{"id": "exact", "distance": 0.0}],
    }
    rows = []
    for index in range(64):
        v = [index / 10, (index * 7 % 17) / 5, (index * 11 % 19) / 7, (index % 3) / 1]
        rows.append(["id-%02d" % index, [fl(x) for x in v]])
    out["full_candidate_adaptive_modes_agree_with_exact_flat_search"] = {
        "cite": "test/vector_adversarial_test.exs:376-421",
        "metric": "l2", "rows": rows, "query": [2.25, 1.5, 0.75, 1.0],
        "limit": 10, "candidates": 64,
        "note": "quantized_search(candidates: 64) ids == flat search ids",
    }
    out["result_values"] = {
        "cite": "lib/vettore_distance.ex:87-102,525-543; test/vector_distance_test.exs:209-217; "
                "test/vector_hardening_test.exs:535-539",
        "table": [
            ["l2", 5.0, "raw", [-5.0, 5.0]],
            ["cosine", 0.25, "raw", [0.25, 0.75]],
            ["l2", 5.0, "similarity", [1.0 / 6.0, 5.0]],
            ["inner_product", 2.0, "similarity", [2.0, -2.0]],
            ["cosine", -1.0, "similarity", [0.0, 2.0]],
            ["jaccard", 1.0, "similarity", [0.5, 1.0]],
            ["negative_inner_product", -1.0, "raw", [1.0, -1.0]],
            ["negative_inner_product", -1.0, "similarity", [1.0, -1.0]],
            ["unknown", 3.0, "unknown", [3.0, None]],
        ],
    }
    out["adapter_validation"] = {
        "cite": "test/vector_hardening_test.exs:497-508; lib/vettore/index/flat.ex:98-112",
        "metric": "l2", "dimensions": 2,
        "invalid_limits": [0, 4294967296],
        "dimension_mismatch_query": [0.0],
    }
    dump("elixir_nif.json", out)


if __name__ == "__main__":
    flat_rs()
    distances_rs()
    search_rs()
    elixir_nif()
    print("wrote fixtures to", HERE)
