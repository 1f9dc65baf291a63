#!/usr/bin/env python3
"""tools/order_sensitivity.py -- how much does the unpinned lane order of wide::f32x8::reduce_add matter?

  python tools/order_sensitivity.py [--rows N] [--dim D] [--queries Q] [--limit K]

The reference's dot / L2 kernels sum each 8-lane chunk with `reduce_add` of the third-party
crate `wide` (distances.rs:236-270), whose lane order depends on the build's target features
and is not on disk here: four orders are implemented, each bit-exact
against the oracle.  This tool runs the SAME headline queries (BASELINE.json: flat cosine
top-10, N = 10 M, d = 768) under all four on one resident corpus and counts what a maintainer
needs to know before pinning the order: how many top-k lists differ from the default order's
-- in their id sets, in their order only, in raw scores only -- and by how many ulps the scores
of the same row differ.  One JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

NAMES = {0: "pair", 1: "avx", 2: "seq", 3: "sse2"}


def ulps(a, b):
    ia = np.frombuffer(np.float32(a).tobytes(), np.int32)[0]
    ib = np.frombuffer(np.float32(b).tobytes(), np.int32)[0]
    return abs(int(ia) - int(ib))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--limit", type=int, default=10)
    a = ap.parse_args()
    import torch
    from vettore_amd import _lib, nifs
    L = _lib.load()
    dev = torch.device("cuda:0")
    x = bench.build_shard(torch, dev, a.rows, a.dim, bench.SEED_CORPUS)
    ref = nifs.flat_new_cosine()
    assert nifs.flat_load_device_matrix(ref, bench.doc_ids(0, a.rows), x.data_ptr(), a.rows, a.dim) == ("ok", ())
    del x
    torch.cuda.empty_cache()
    qs = bench.normalized_queries(a.queries, a.dim, bench.SEED_QUERY)
    results = {}
    for order in (3, 0, 1, 2):
        assert nifs.flat_set_reduce_order(ref, order) == "ok"
        lists = []
        for q in qs:
            h = C.c_void_p()
            assert L.vt_flat_search(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), a.dim, a.limit, C.byref(h)) == 0
            lists.append(nifs._take_hits(h))
        results[order] = lists
    base = results[3]
    table = {}
    for order in (0, 1, 2):
        diff_ids = diff_order = diff_raw_only = 0
        max_ulps = 0
        ulp_hist = {}
        boundary_swaps = 0
        for bl, ol in zip(base, results[order]):
            b_ids, o_ids = [h[0] for h in bl], [h[0] for h in ol]
            if set(b_ids) != set(o_ids):
                diff_ids += 1
                # did only the last place change hands?
                if set(b_ids[:-1]) == set(o_ids[:-1]):
                    boundary_swaps += 1
            elif b_ids != o_ids:
                diff_order += 1
            elif [np.float32(h[1]).tobytes() for h in bl] != [np.float32(h[1]).tobytes() for h in ol]:
                diff_raw_only += 1
            o_raw = {h[0]: h[1] for h in ol}
            for h in bl:
                if h[0] in o_raw:
                    u = ulps(h[1], o_raw[h[0]])
                    max_ulps = max(max_ulps, u)
                    ulp_hist[u] = ulp_hist.get(u, 0) + 1
        table[NAMES[order]] = {"lists_with_other_ids": diff_ids, "of_those_only_the_last_place": boundary_swaps,
                               "lists_same_ids_other_order": diff_order, "lists_same_order_other_raw_bits": diff_raw_only,
                               "max_ulps_same_row": max_ulps,
                               "ulps_histogram_same_row": {str(k): v for k, v in sorted(ulp_hist.items())}}
    print(json.dumps({"rows": a.rows, "dim": a.dim, "queries": a.queries, "limit": a.limit, "baseline_order": "sse2",
                      "against": table}))


if __name__ == "__main__":
    main()
