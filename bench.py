#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: queries/sec (+ achieved HBM GB/s) of flat
cosine top-10 search, N=10M, d=768, single query in flight, on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--rows N] [--dim D]

A step = one `flat_search` call (one query scanned against the whole corpus)
through the C ABI of libvettore_hip.so: query H2D, scan + fused top-k kernel,
merge kernel, result D2H.  The corpus is resident in HBM before timing starts.

Setup ends with a pause (r06, --release-wait, 2.5 s): the build hands its 30-GB source tensor back to the driver, and for
~1.2 s after such a release every scan of the card runs 2.5 % slower -- the driver's 20 steps (0.1 s) used to sit inside
that stretch (tools/ramp_probe.py, profiles/r06/ramp_probe.jsonl).  Then W warm-up steps and EXACTLY K timed steps
between barrier + synchronize, as the contract says.
At N = 8 the line also carries BASELINE configs[3] (L2, rows over eight GPUs) as `side.config4`.

N > 1: the SAME 10M-row corpus is sharded across N GPUs (strong scaling); every
query runs on every shard and the per-shard top-k lists meet in one all-gather
(the path's only exchange step), then merge by (rank key, id bytes).  Two ways
to get there, same library underneath:
  * `python bench.py --gpus N` (no launcher): ONE process, ONE index handle
    over N devices (vt_flat_new_sharded) -- the shape of the reference's NIF
    resource (nifs.rs:254-257); the library runs ncclAllGather itself
    (`rccl_ranks` = ncclCommCount of its communicator);
  * under torch.distributed.run (WORLD_SIZE set): one rank per GPU, each with
    its own one-device index, torch.distributed (nccl = RCCL) carries the
    all-gather (vettore_amd/sharded.py); `rccl_ranks` = dist.get_world_size().

At N = 1 the JSON line also carries `side`: short legs of the other BASELINE
configs on the same box (config 2: N=1M single query; config 3: dot, batches of
256 on the FP32 matrix cores; config 5: quantized search; funnel search), each
with its own roofline figure, and `callers`: 8 and 64 threads searching the
headline index at the same time.  The headline fields are not affected.

Synthetic data: iid uniform(-1,1) coordinates, rows
L2-normalised, 1% verbatim duplicate rows, ids "doc-<i>", seeds 20260721/22.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PMC_SIDE_SOURCE = "profiles/pmc_side.json (rocprofv3 --pmc FETCH_SIZE passes of this script's --mode legs, tools/refresh_profiles.sh; not this run)"
PMC_SOURCE = "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this script, tools/refresh_profiles.sh; not this run)"
SEED_CORPUS, SEED_QUERY = 20260721, 20260722


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 1000: 4.6 s of scans at the headline size)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps first (default 50)")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--limit", type=int, default=10)
    ap.add_argument("--metric", default="cosine",
                    choices=["cosine", "l2", "l2_squared", "inner_product", "negative_inner_product", "manhattan", "chebyshev"],
                    help="single mode: the index metric (default: the headline's cosine; BASELINE configs[3] is "
                         "`--metric l2 --rows 40000000 --gpus 8`)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--mode", choices=["single", "batch", "quantized", "funnel"], default="single",
                    help="single: BASELINE.json metric (default); batch: configs[2] (dot, 256-query batches, "
                         "FP32 MFMA); quantized: configs[4] (sign-bit Hamming pass + exact rerank)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the all_gather exchange even on one rank (prices the multi-GPU merge step)")
    ap.add_argument("--stages", default="128", help="funnel mode: prefix lengths (collection.ex:660-672 default min(d,128))")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--nominate", choices=["bf16", "f32"], default="bf16",
                    help="batch mode: which matrix-core pass names the candidates (K2b bf16, HBM-bound / K2 FP32 MFMA)")
    ap.add_argument("--candidates", type=int, default=100)
    ap.add_argument("--reduce-order", choices=["pair", "avx", "seq", "sse2"], default="sse2",
                    help="lane order of wide::f32x8::reduce_add the kernels reproduce (include/vettore_flat.h)")
    ap.add_argument("--no-side", action="store_true", help="skip the side legs (configs 2, 3, 5, funnel) at N=1")
    ap.add_argument("--devices", default=None,
                    help="one process, N devices: HIP ordinals of the shards (default 0..N-1); an ordinal may repeat, "
                         "which puts several shards on one GPU (how a one-GPU box exercises --gpus 2)")
    ap.add_argument("--exchange", choices=["auto", "rccl", "host"], default="auto",
                    help="how the shards' lists meet: one process, N devices: the library's RCCL all-gather or its host-mapped "
                         "lists; under torch.distributed.run: backend nccl (RCCL) with the device-side merge, or gloo with "
                         "64-byte records (auto = RCCL first, see --supervise)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = --rows is the whole corpus, sharded over the GPUs (the north star's curve); "
                         "weak = --rows per GPU (N x --rows in all), reported as \"scaling\": \"weak\"")
    ap.add_argument("--shadow", choices=["auto", "off"], default="auto",
                    help="batch mode: let the bf16 pass keep a bf16 shadow of the rows (K2s) or stream the f32 rows (K2b)")
    ap.add_argument("--config4-rows", type=int, default=5_000_000,
                    help="--gpus 8 only: rows PER GPU of the side.config4 leg (BASELINE configs[3]: L2, N = 40 M over 8 GPUs)")
    ap.add_argument("--config4-anyway", action="store_true", help=argparse.SUPPRESS)   # (tests: the leg at any width, over any exchange)
    ap.add_argument("--release-wait", type=float, default=2.5,
                    help="seconds the card is left alone after the build has handed its source tensor back to the driver (0: none, "
                         "rounds 1-5): for ~1.2 s after a 30-GB release every scan runs 2.5 %% slower (tools/ramp_probe.py)")
    ap.add_argument("--supervise", action="store_true",
                    help="run the measurement in a child process and, should it fail or hang, once more over the host exchange "
                         "(always on for N > 1; this flag switches it on at N = 1, with --exchange rccl|host)")
    ap.add_argument("--debug-set", action="append", default=[], metavar="NAME=VALUE",
                    help="a library setting for this run (vt_debug_set, include/vettore_flat.h), e.g. force_batch_mfma=1 on a small corpus")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)   # (the supervised measurement itself)
    ap.add_argument("--child-timeout", type=float, default=480.0, help="seconds a supervised measurement may take")
    ap.add_argument("--exchange-note", default=None, help=argparse.SUPPRESS)  # (why this child runs over the host exchange)
    return ap.parse_args()


ORDER_CODE = {"pair": 0, "avx": 1, "seq": 2, "sse2": 3}


def doc_ids(start, count, idx=None):
    """ids "doc-<i>" for i in [start+1, start+count] (or for the given i's) as (bytes, offsets)."""
    if idx is None:
        idx = np.arange(start + 1, start + count + 1, dtype=np.int64)
    count = len(idx)
    digits = np.floor(np.log10(idx)).astype(np.int64) + 1
    off = np.zeros(count + 1, dtype=np.uintp)
    off[1:] = np.cumsum(digits + 4)
    blob = b"".join([b"doc-%d" % i for i in idx.tolist()])
    assert len(blob) == int(off[-1])
    return blob, off


def build_shard(torch, device, rows, dim, seed, chunk=1 << 20, normalize=True):
    """uniform(-1,1) rows, L2-normalised (cosine collections: collection.ex:1317-1319; the other
    metrics store what they are given), 1% verbatim duplicates, generated in HBM."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.empty((rows, dim), dtype=torch.float32, device=device)
    for s in range(0, rows, chunk):
        e = min(rows, s + chunk)
        blk = torch.rand((e - s, dim), generator=g, device=device, dtype=torch.float32) * 2.0 - 1.0
        if normalize:
            blk /= torch.linalg.vector_norm(blk.double(), dim=1, keepdim=True).float()
        x[s:e] = blk
        del blk
    ndup = rows // 100
    if ndup:
        src = torch.randint(0, rows, (ndup,), generator=g, device=device)
        dst = torch.randint(0, rows, (ndup,), generator=g, device=device)
        x[dst] = x[src]
    torch.cuda.synchronize()
    return x


def release_pause(a):
    """After a build has handed its source tensor back to the driver (`del x; empty_cache()`): for ~1.2 s after a 30-GB release
    every scan of the card runs 2.5 % slower (measure(); tools/ramp_probe.py).  Setup, never inside a timed region."""
    if getattr(a, "release_wait", 0) > 0:
        time.sleep(a.release_wait)


def cpu_baseline(dim, limit, budget_s):
    """The CPU beside it: the oracle -- the reference's algorithm restated in C, pinned to
    the reference's own test vectors -- timed on this box's host cores on a bounded sample of the same
    workload.  The reference cannot be built here (Rust + un-vendored crates), so kind = "port".  Every
    flavour SURVEY 8d names is timed (tools/cpu_variants.py, one child process per build):
      * shape: the reference's own (hash map of separately allocated rows, id clone per row, bounded
        heap: flat.rs:96-124) and a contiguous row matrix ("best-effort CPU": what the reference's
        layout costs it stays visible);
      * build: -O3 for baseline x86-64 (what a precompiled release NIF targets) and -march=native
        (Taskfile.yml:12);
      * 1 thread (a reference search is single-threaded) and T = host cores concurrent readers (how
        the reference gets throughput: dirty schedulers under RwLock::read, nifs.rs:304-308).
    The sample is >= 4 GB of rows, far beyond the last-level cache (r03 timed 614 MB, which flattered
    the CPU on a 256-core box).  `value` is the reference-shaped, x86-64, T-reader
    figure: what a user of the published NIF gets from this box."""
    import subprocess
    rows = max(200_000, int(4.0e9 / (dim * 4)))
    legs = 8
    per_leg = max(1.0, budget_s / legs)
    env = dict(os.environ, ROWS=str(rows), SECONDS_PER_LEG="%.2f" % per_leg)
    t0 = time.perf_counter()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_variants.py")], env=env, capture_output=True, text=True,
                         timeout=600)
    if res.returncode != 0:
        raise RuntimeError("tools/cpu_variants.py failed: " + res.stderr[-2000:])
    variants = [json.loads(line) for line in res.stdout.splitlines() if line.startswith("{")]
    for v in variants:
        v["shape"] = "reference" if v["shape"].startswith("reference") else "contiguous"
    def pick(build, shape, many):
        for v in variants:
            if v["build"] == build and v["shape"] == shape and (v["threads"] > 1) == many:
                return v
        return None
    head, one = pick("x86-64", "reference", True), pick("x86-64", "reference", False)
    return {"rows_per_s_T": head["rows_per_s"], "rows_per_s_1": one["rows_per_s"], "threads": head["threads"], "sample_rows": rows,
            "sample_bytes": rows * dim * 4, "seconds": time.perf_counter() - t0, "seconds_per_leg": per_leg, "variants": variants}


_READ_PEAK = {}


def measured_read_peak(device=0):
    """GB/s of the plainest read-only streaming kernel, measured HERE, in this run, on this box
    (vt_device_read_peak: 8 GiB scratch buffer, best of 5 passes; VERDICT r2 weak #9 -- it used to
    be quoted from a round-1 file).  None if the measurement fails."""
    if device not in _READ_PEAK:
        from vettore_amd import nifs
        res = nifs.device_read_peak(device, 8 << 30, 5)
        _READ_PEAK[device] = res[1] if res[0] == "ok" else None
    return _READ_PEAK[device]


def pmc_traffic(rows, dim):
    """HBM bytes per scan launch from the committed rocprofv3 PMC passes
    (profiles/pmc_latest.json: FETCH_SIZE and WRITE_SIZE collected in separate
    --pmc runs of this script, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950).  bench.py cannot read counters itself, so the figure
    is reported only when the profile was taken on the same workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            p = json.load(f)
        if p.get("rows") == rows and p.get("dim") == dim:
            return p["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def pmc_side_traffic(kernel, nbytes_or_flops_rows, dim):
    """HBM bytes per launch of a side leg's dominant kernel from the committed PMC passes
    (profiles/pmc_side.json, written by tools/refresh_profiles.sh: FETCH_SIZE x 2 + WRITE_SIZE,
    as for the headline), when they were taken on the same shape; else None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_side.json")) as f:
            p = json.load(f).get(kernel)
        if p and p.get("rows") == nbytes_or_flops_rows and p.get("dim") == dim:
            return p["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def verify_against_oracle(nifs, dim, order_code, limit=10):
    """The checker of the side legs (they used to compare a call with a repeat of
    itself).  A small resident side index -- 20 000 x dim cosine rows, enough for every leg to take
    the kernels it takes at full size (K1, the Hamming histogram pass, the f64 prefix scan, K2b) --
    answers one query per entry point, and each answer must equal the CPU oracle's composition of
    the reference's functions bit for bit.  The oracle is test infrastructure: it checks, outside
    every timed region; nothing measured runs through it."""
    import oracle
    oracle.build()
    oracle.set_reduce_order(order_code)
    try:
        rng = np.random.default_rng(SEED_CORPUS + 77)
        n = 20_000
        x = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
        x[500:540] = x[500]                                     # a block of identical rows: the id bytes decide
        x = np.stack([oracle.normalize_l2(r) for r in x])
        ids = [b"doc-%d" % (i + 1) for i in range(n)]
        ref = nifs.flat_new_cosine()
        nifs.flat_set_reduce_order(ref, order_code)
        assert nifs.flat_load_matrix(ref, ids, x) == ("ok", ())
        packed = oracle.pack_ids(ids)
        q = oracle.normalize_l2(rng.uniform(-1, 1, dim).astype(np.float32))
        bits = lambda hits: [(h[0], np.float32(h[1]).tobytes()) for h in hits]  # noqa: E731
        rows = [(ids[i], x[i]) for i in range(n)]
        by_id = dict(rows)
        out = {}
        # flat_search (flat.rs:96-124)
        out["single"] = bits(nifs.flat_search(ref, q, limit)[1]) == bits(oracle.matrix_search(2, x, packed, q, limit))
        out["single_on_a_tie_block"] = bits(nifs.flat_search(ref, x[500], limit)[1]) == bits(oracle.matrix_search(2, x, packed, x[500], limit))
        # quantized_search (collection.ex:276-295): binary_top_k over sign bits, then vector_top_k
        words = (dim + 63) // 64
        signs = np.zeros((n, words * 64), dtype=np.uint8)
        signs[:, :dim] = x >= 0
        packed_bits = np.packbits(signs, axis=1, bitorder="little").view(np.uint64)
        cands = oracle.binary_top_k([(ids[i], packed_bits[i]) for i in range(n)], oracle.compress_sign_bits(q), dim, 100)
        want = oracle.vector_top_k([(c, by_id[c]) for c, _ in cands], q, 2, dim, limit)
        out["quantized"] = bits(nifs.flat_quantized_search(ref, q, 100, limit)[1]) == bits(want)
        # funnel_search (collection.ex:245-260, :674-691): vector_top_k on the prefix, then on the full rows
        stage = min(dim, 128)
        kept = oracle.vector_top_k(rows, q, 2, stage, 100)
        want = oracle.vector_top_k([(i, by_id[i]) for i, _ in kept], q, 2, dim, limit)
        out["funnel"] = bits(nifs.flat_funnel_search(ref, q, [stage], 100, limit)[1]) == bits(want)
        # a query batch through the matrix cores (both nominations) == the oracle's single searches
        qs = np.stack([oracle.normalize_l2(v) for v in rng.uniform(-1, 1, (16, dim)).astype(np.float32)])
        with nifs.debug_setting("force_batch_mfma", 1):   # (the cost model would send a corpus this small to single scans)
            for name, mode in (("batch_bf16", 2), ("batch_f32", 1)):
                nifs.flat_set_batch_nominate(ref, mode)
                got = nifs.flat_search_batch(ref, qs, limit)[1]
                out[name] = all(bits(got[i]) == bits(oracle.matrix_search(2, x, packed, qs[i], limit)) for i in range(16))
        # flat_search under float hamming / jaccard from the non-zero-bit column (K4), a batch in K4p sweeps
        xs = (rng.uniform(-1, 1, (n, dim)) * (rng.uniform(0, 1, (n, dim)) < 0.5)).astype(np.float32)
        qsp = (rng.uniform(-1, 1, (16, dim)) * (rng.uniform(0, 1, (16, dim)) < 0.5)).astype(np.float32)
        for name, code in (("pattern_hamming", 7), ("pattern_jaccard", 8)):
            refp = nifs._flat_new(code)
            assert nifs.flat_load_matrix(refp, ids, xs) == ("ok", ())
            nifs.flat_set_profiling(refp, True)
            got1 = nifs.flat_search(refp, qsp[0], limit)[1]
            gotb = nifs.flat_search_batch(refp, qsp, limit)[1]
            prof = nifs.flat_get_profile(refp, reset=True)
            out[name] = (bits(got1) == bits(oracle.matrix_search(code, xs, packed, qsp[0], limit)) and
                         all(bits(gotb[i]) == bits(oracle.matrix_search(code, xs, packed, qsp[i], limit)) for i in range(16)) and
                         prof["scan_launches"] == 0 and prof["hamming_launches"] >= 2)
            del refp
        assert all(out.values()), out
        return {"checker": "oracle/ (CPU restatement of flat.rs / search.rs / distances.rs)", "side_index_rows": n, "equal_bit_for_bit": out}
    finally:
        oracle.set_reduce_order(oracle.DEFAULT_ORDER)


def hits_of(L, handle_ptr):
    """[(id, raw bits)] of a vt_hits handle (freed)."""
    from vettore_amd import nifs
    return [(h[0], np.float32(h[1]).tobytes()) for h in nifs._take_hits(handle_ptr)]


def leg(a, L, nifs, ref, mode, qs, steps, warmup, per=1, stages=(128,), candidates=100, limit=10, kernel=None):
    """Times `steps` calls of one entry point on the resident index `ref` (after `warmup`
    untimed ones) and returns {ms_per_step, value, roofline...} from the library's HIP-event
    profile.  At full size a result per leg is compared with the single-query path (batch) or
    with a second run of itself (the other modes); the entry point itself is checked against the
    CPU oracle on a small side index by verify_against_oracle(), once per run."""
    import torch
    dim = qs.shape[1]
    outs = (C.c_void_p * per)()
    st_arr = (C.c_size_t * len(stages))(*stages)

    def call(i, keep=False):
        q = qs[i * per:(i + 1) * per]
        qp = q.ctypes.data_as(C.POINTER(C.c_float))
        if mode == "batch":
            assert L.vt_flat_search_batch(ref.handle, qp, per, dim, limit, outs) == 0
            if not keep:
                # (one call hands the lists back: 256 separate frees through ctypes were 0.2 ms of Python per 4.4-ms step)
                L.vt_hits_free_many(outs, per)
                return None
            res = [C.c_void_p(outs[j]) for j in range(per)]
        else:
            h = C.c_void_p()
            if mode in ("single", "pattern"):
                st = L.vt_flat_search(ref.handle, qp, dim, limit, C.byref(h))
            elif mode == "funnel":
                st = L.vt_flat_funnel_search(ref.handle, qp, dim, st_arr, len(stages), candidates, limit, C.byref(h))
            else:
                st = L.vt_flat_quantized_search(ref.handle, qp, dim, candidates, limit, C.byref(h))
            assert st == 0, st
            res = [h]
        if keep:
            return [hits_of(L, r) for r in res]
        for r in res:
            L.vt_hits_free(r)
        return None

    for i in range(warmup):
        call(i)
    # first the steps with the library's HIP-event bookkeeping, for the dominant kernel's own
    # duration (this pass also settles clocks: the first ~1000 short calls after a load run
    # 5-10 % slow) ...
    nifs.flat_set_profiling(ref, True)
    nifs.flat_get_profile(ref, reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        call(i)
    torch.cuda.synchronize()
    dt_profiled = time.perf_counter() - t0
    p = nifs.flat_get_profile(ref, reset=True)
    nifs.flat_set_profiling(ref, False)
    # ... then the same steps end to end WITHOUT it (two event records per call put ~6-10 us of
    # barrier packets into a 0.2 ms chain)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        call(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # verification outside the timed region
    last = call(warmup + steps - 1, keep=True)
    if mode == "batch":
        for j in (0, per - 1):
            h = C.c_void_p()
            q = qs[(warmup + steps - 1) * per + j]
            assert L.vt_flat_search(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), dim, limit, C.byref(h)) == 0
            assert hits_of(L, h) == last[j], "batched result differs from the single-query path"
    else:
        assert call(warmup + steps - 1, keep=True) == last and len(last[0]) == limit
    out = {"ms_per_step": dt / steps * 1e3, "value": steps * per / dt, "unit": "queries/s", "steps": steps, "warmup": warmup,
           "ms_per_step_with_event_timing": dt_profiled / steps * 1e3, "verified": True}
    if mode == "batch":
        out["fallback_queries"] = p["batch_fallbacks"]
        out["roofline"] = batch_roofline(p, len(ref), dim)
        if p["nominate_launches"]:
            assert p["batch_launches"] == 0
            launches = p["nominate_launches"]
            out["second_passes"] = p["nominate_second_passes"]
            out["candidates_per_query"] = p["nominate_candidates"] / max(1, p["nominate_queries"])
            out["bf16_shadow"] = nifs.flat_batch_shadow(ref)
            # (all the passes of the timed steps over the time of those steps: a step of 16 x 256 is sixteen passes)
            out["end_to_end_frac"] = p["nominate_bytes"] / dt / 1e9 / HBM_PEAK_GBS
            out["passes_per_step"] = launches / steps
    else:
        key = {"single": "scan", "funnel": "prefix", "quantized": "hamming", "pattern": "hamming"}[mode]
        kern = {"single": "scan_topk_kernel", "funnel": "cosine_scan_kernel", "quantized": "hamming_dist_kernel",
                "pattern": "hamming_topk_kernel"}[mode]
        kern = kernel or kern
        if mode == "pattern" and len(ref) >= 16384:  # (every search a pass over the non-zero-bit column, none a scan of the rows)
            assert p["scan_launches"] == 0 and p["hamming_launches"] >= steps, p
        launches = max(1, p[key + "_launches"])
        ms = p[key + "_ms"] / launches
        gbs = p[key + "_bytes"] / launches / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out["roofline"] = {"bound": "hbm", "kernel": kern, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": gbs / HBM_PEAK_GBS,
                           "traffic": pmc_side_traffic(kern, len(ref), dim) if mode not in ("single", "pattern") else None,
                           "avg_launch_ms": ms,
                           "algorithmic_bytes_per_launch": p[key + "_bytes"] / launches}
        # end to end against the same algorithmic bytes (launch chain + host waits included)
        out["end_to_end_frac"] = p[key + "_bytes"] / launches / (dt / steps) / 1e9 / HBM_PEAK_GBS
    return out


def leg_single_via_shadow(a, L, nifs, ref, qs, steps, warmup, limit):
    """A side leg, never the headline: lone flat_search calls on the headline index with
    vt_flat_set_single_nominate on -- each answered like a batch of one: the bf16 pass over the bf16
    shadow of the rows (N x d x 2 bytes, half of what the exact scan reads) nominates a few hundred
    rows, the exact kernel re-scores them, the bound certifies the top k; a query it cannot certify
    takes the exact scan.  Every timed answer is compared with the plain scan's (taken first)."""
    import torch
    dim = qs.shape[1]
    h = C.c_void_p()

    def call(i, keep=False):
        st = L.vt_flat_search(ref.handle, qs[i].ctypes.data_as(C.POINTER(C.c_float)), dim, limit, C.byref(h))
        assert st == 0, st
        if keep:
            return hits_of(L, h)
        L.vt_hits_free(h)
        return None

    plain = [call(i, keep=True) for i in range(warmup, warmup + steps)]
    assert nifs.flat_set_single_nominate(ref, True) == ("ok", ())
    try:
        t0 = time.perf_counter()
        call(0)   # builds the row norms and the shadow: setup
        setup_s = time.perf_counter() - t0
        for i in range(warmup):
            call(i)
        nifs.flat_set_profiling(ref, True)
        nifs.flat_get_profile(ref, reset=True)
        got = [call(i, keep=True) for i in range(warmup, warmup + steps)]   # (first pass: verification and the kernel's own time)
        p = nifs.flat_get_profile(ref, reset=True)
        nifs.flat_set_profiling(ref, False)
        assert got == plain, "a search through the shadow differs from the exact scan"
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            call(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        assert nifs.flat_set_single_nominate(ref, False) == ("ok", ())
    out = {"ms_per_step": dt / steps * 1e3, "value": steps / dt, "unit": "queries/s", "steps": steps, "warmup": warmup, "verified": True,
           "certified": p["nominate_queries"] - p["batch_fallbacks"], "took_the_exact_scan": p["batch_fallbacks"],
           "shadow_setup_s": round(setup_s, 2), "roofline": batch_roofline(p, len(ref), dim),
           "candidates_per_query": p["nominate_candidates"] / max(1, p["nominate_queries"])}
    out["end_to_end_frac"] = p["nominate_bytes"] / max(1, p["nominate_launches"]) / (dt / steps) / 1e9 / HBM_PEAK_GBS
    return out


def _lib_consts():
    from vettore_amd import _lib
    return _lib


def normalized_queries(n, dim, seed):
    qs = np.random.default_rng(seed).uniform(-1, 1, size=(n, dim)).astype(np.float32)
    qs /= np.linalg.norm(qs.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    return qs


def run_side_mode(a, torch, nifs, device):
    """`--mode batch|quantized|funnel`: that config alone, its own JSON line (profiling runs)."""
    from vettore_amd import _lib
    L = _lib.load()
    batch = a.mode == "batch"
    x = build_shard(torch, device, a.rows, a.dim, SEED_CORPUS)
    if batch:  # configs[2]: metric :dot -> :inner_product, no normalisation (collection.ex:1302, :1319)
        x.mul_(torch.empty((a.rows, 1), device=device).uniform_(8.0, 24.0))
        ref = nifs.flat_new_inner_product()
    else:
        ref = nifs.flat_new_cosine()
    nifs.flat_set_reduce_order(ref, ORDER_CODE[a.reduce_order])
    assert nifs.flat_set_batch_nominate(ref, _lib.NOMINATE_BF16 if a.nominate == "bf16" else _lib.NOMINATE_F32) == "ok"
    if a.shadow == "off":
        assert nifs.flat_set_batch_shadow(ref, _lib.SHADOW_OFF) == ("ok", ())
    assert nifs.flat_load_device_matrix(ref, doc_ids(0, a.rows), x.data_ptr(), a.rows, a.dim) == ("ok", ())
    del x
    torch.cuda.empty_cache()
    release_pause(a)
    per = a.batch if batch else 1
    nq = (a.steps + a.warmup) * per
    if batch:
        qs = np.random.default_rng(SEED_QUERY).uniform(-1, 1, size=(nq, a.dim)).astype(np.float32)
    else:
        qs = normalized_queries(nq, a.dim, SEED_QUERY)
    stage_list = [int(v) for v in a.stages.split(",")]
    r = leg(a, L, nifs, ref, a.mode, qs, a.steps, a.warmup, per=per, stages=stage_list, candidates=a.candidates, limit=a.limit)
    out = {"value": r["value"], "unit": "queries/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": {"batch": "f32" if a.nominate == "f32" else "f32 (exact rescoring; bf16 nomination)", "funnel": "f64",
                     "quantized": "u64"}[a.mode], "data": "synthetic",
           "roofline": r["roofline"]}
    if batch:
        out["metric"] = "queries/sec, flat dot top-%d, N=%d d=%d, batch=%d" % (a.limit, a.rows, a.dim, per)
        out["config"] = {"workload": "index: :flat, metric: :dot, d=%d, N=%d, batch=%d queries (%s MFMA Q x D^T + exact rescoring)"
                         % (a.dim, a.rows, per, a.nominate), "fallback_queries": r["fallback_queries"],
                         "bf16_shadow": r.get("bf16_shadow")}
        out["ms_per_step_with_event_timing"] = r["ms_per_step_with_event_timing"]
    elif a.mode == "funnel":
        out["metric"] = "queries/sec, funnel_search (f64 cosine on prefix %s, keep %d, exact rerank top-%d), N=%d d=%d" % (
            a.stages, a.candidates, a.limit, a.rows, a.dim)
        out["config"] = {"workload": "funnel_search stages=[%s] candidates=%d limit=%d, d=%d, N=%d" % (
            a.stages, a.candidates, a.limit, a.dim, a.rows)}
    else:
        out["metric"] = "queries/sec, quantized_search (sign-bit Hamming top-%d + exact cosine rerank top-%d), N=%d d=%d" % (
            a.candidates, a.limit, a.rows, a.dim)
        out["config"] = {"workload": "quantized_search candidates=%d limit=%d, d=%d, N=%d" % (a.candidates, a.limit, a.dim, a.rows)}
    out["config"]["reduce_order"] = a.reduce_order
    print(json.dumps(out))


_CALLERS_LIB = None


def native_callers(a, L, nifs, ref, threads, seconds, qs, quantized, funnel):
    """The callers as NATIVE threads (vettore_amd/lib/libvt_callers.so, tools/callers_native.cpp): Python
    threads must win the interpreter lock before they can call again and return to the handle in a
    trickle, which hides how the library lets concurrent callers share a pass.  Every 8th
    answer of a thread is compared with the same query's answer alone.  None when the helper is not
    built: the caller falls back to Python threads."""
    global _CALLERS_LIB
    if _CALLERS_LIB is None:
        path = os.path.join(ROOT, "vettore_amd", "lib", "libvt_callers.so")
        try:
            lib = C.CDLL(path) if os.path.exists(path) else False
        except OSError:
            lib = False
        if lib:
            lib.vt_callers_run.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                           C.c_size_t, C.c_size_t, C.c_int, C.c_double, C.c_int,
                                           C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
            lib.vt_callers_run.restype = C.c_int
        _CALLERS_LIB = lib
    if not _CALLERS_LIB:
        return None
    q = np.ascontiguousarray(qs, dtype=np.float32)
    kind, param, cand = (2, funnel, 100) if funnel else (1, 0, quantized) if quantized else (0, 0, 0)
    n, wrong, failed = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
    b0 = nifs.flat_coalesce_stats(ref)
    t0 = time.perf_counter()
    rc = _CALLERS_LIB.vt_callers_run(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), len(q), a.dim, a.limit, kind, param, cand,
                                     threads, seconds, 8, C.byref(n), C.byref(wrong), C.byref(failed))
    dt = time.perf_counter() - t0
    if rc != 0:
        return None
    b1 = nifs.flat_coalesce_stats(ref)
    # (dt includes the answers taken alone before the threads start: the helper reports its own window)
    return {"threads": threads, "callers": "native threads", "value": n.value / seconds, "unit": "queries/s", "seconds": round(seconds, 2),
            "wall_seconds": round(dt, 2), "searches": n.value, "batches": b1[0] - b0[0], "searches_in_batches": b1[1] - b0[1],
            "verified": wrong.value == 0 and failed.value == 0, "mean_latency_ms": seconds * threads / max(1, n.value) * 1e3}


def concurrent_callers(a, L, nifs, ref, threads, seconds, quantized=0, funnel=0):
    """`threads` callers searching ONE handle at the same time (what BEAM dirty schedulers do under
    the reference's read lock, nifs.rs:297-309): the library lets searches that meet travel as one
    batch.  Every answer is compared with the one the same query gets alone.  quantized = c > 0:
    the callers run quantized_search(candidates: c) instead (collection.ex:276-295); funnel = p > 0:
    funnel_search(stages: [p], candidates: 100) (collection.ex:245-260)."""
    import threading
    qs = normalized_queries(64, a.dim, SEED_QUERY + 9)
    native = native_callers(a, L, nifs, ref, threads, seconds, qs, quantized, funnel)
    if native is not None:
        return native
    hp = C.c_void_p()
    stages = (C.c_size_t * 1)(funnel)

    def one(q, h):
        qp = q.ctypes.data_as(C.POINTER(C.c_float))
        if funnel:
            return L.vt_flat_funnel_search(ref.handle, qp, a.dim, stages, 1, 100, a.limit, C.byref(h))
        if quantized:
            return L.vt_flat_quantized_search(ref.handle, qp, a.dim, quantized, a.limit, C.byref(h))
        return L.vt_flat_search(ref.handle, qp, a.dim, a.limit, C.byref(h))

    alone = []
    for q in qs:
        assert one(q, hp) == 0
        alone.append(hits_of(L, hp))
    stop, counts, wrong = threading.Event(), [0] * threads, []

    def worker(t):
        h = C.c_void_p()
        i = t
        while not stop.is_set():
            j = i % len(qs)
            assert one(qs[j], h) == 0
            if counts[t] % 8 == 0:
                if hits_of(L, h) != alone[j]:
                    wrong.append((t, j))
            else:
                L.vt_hits_free(h)
            counts[t] += 1
            i += threads

    b0 = nifs.flat_coalesce_stats(ref)
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    time.sleep(seconds)
    stop.set()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    b1 = nifs.flat_coalesce_stats(ref)
    total = sum(counts)
    return {"threads": threads, "callers": "python threads", "value": total / dt, "unit": "queries/s", "seconds": round(dt, 2), "searches": total,
            "batches": b1[0] - b0[0], "searches_in_batches": b1[1] - b0[1], "verified": not wrong,
            "mean_latency_ms": dt * threads / max(1, total) * 1e3}


def side_legs(a, torch, nifs, L, device, main_ref):
    """The other BASELINE configs, briefly, on the same box.  `main_ref` is
    the resident N-row cosine index of the headline leg."""
    side = {}
    t0 = time.perf_counter()
    # every leg's entry point, once, against the oracle (a deterministic wrong answer cannot post a number)
    side["verified_against_oracle"] = verify_against_oracle(nifs, a.dim, ORDER_CODE[a.reduce_order], a.limit)
    # config 5: quantized_search on the resident cosine corpus (sign bits are built by the first call)
    qs = normalized_queries(330, a.dim, SEED_QUERY + 5)
    side["config5"] = dict(leg(a, L, nifs, main_ref, "quantized", qs, 300, 30, candidates=100),
                           workload="quantized_search candidates=100 limit=10, d=%d, N=%d" % (a.dim, a.rows), dtype="u64",
                           note="end to end = upload + five launches + one wait around a distance pass that sits at the small-transfer "
                                "floor (a bare read of the same 0.96 GB: 0.153 ms); fusing the chain was measured and lost to L2 "
                                "write-back (DESIGN_APPENDIX A.12)")
    # ... and as ONE quantized_search_batch call of 64: groups of eight share a sweep of the sign bits, the groups of
    # the call are queued on two streams
    outs64 = (C.c_void_p * 64)()
    qb = np.ascontiguousarray(qs[:64])
    times = []
    for rep in range(6):
        t1 = time.perf_counter()
        assert L.vt_flat_quantized_search_batch(main_ref.handle, qb.ctypes.data_as(C.POINTER(C.c_float)), 64, a.dim, 100, a.limit, outs64) == 0
        times.append(time.perf_counter() - t1)
        got = [hits_of(L, C.c_void_p(outs64[j])) for j in range(64)]
    for j in range(0, 64, 9):
        h = C.c_void_p()
        assert L.vt_flat_quantized_search(main_ref.handle, qb[j].ctypes.data_as(C.POINTER(C.c_float)), a.dim, 100, a.limit, C.byref(h)) == 0
        assert hits_of(L, h) == got[j], "batched quantized search differs from the single call"
    side["config5"]["batch64"] = {"ms": min(times[1:]) * 1e3, "queries_per_s": 64 / min(times[1:]), "kernel": "hamming_dist_multi_kernel",
                                  "verified": True}
    # funnel_search, prefix 128
    qs = normalized_queries(110, a.dim, SEED_QUERY + 6)
    side["funnel"] = dict(leg(a, L, nifs, main_ref, "funnel", qs, 100, 10, stages=(min(a.dim, 128),), candidates=100),
                          workload="funnel_search stages=[%d] candidates=100 limit=10, d=%d, N=%d" % (min(a.dim, 128), a.dim, a.rows),
                          dtype="f64")
    # lone searches through the bf16 shadow of the headline index (opt-in; a side leg, never the headline)
    qs = normalized_queries(230, a.dim, SEED_QUERY + 11)
    side["single_via_shadow"] = dict(leg_single_via_shadow(a, L, nifs, main_ref, qs, 200, 30, a.limit),
                                     workload="index: :flat, metric: :cosine, d=%d, N=%d, single query, vt_flat_set_single_nominate on: "
                                              "bf16 pass over the bf16 shadow + exact rescoring + certified bound" % (a.dim, a.rows),
                                     dtype="f32 (exact rescoring; bf16 nomination)")
    # many callers on the one handle (the headline has one query in flight)
    side["callers"] = {"workload": "index: :flat, metric: :cosine, d=%d, N=%d, limit=%d, T threads calling flat_search on one handle"
                                   % (a.dim, a.rows, a.limit),
                       "runs": [concurrent_callers(a, L, nifs, main_ref, t, 1.5) for t in (8, 64)]}
    # ... and many quantized_search callers: up to eight share a sweep of the sign-bit matrix
    side["callers_quantized"] = {"workload": "quantized_search candidates=100 limit=%d, d=%d, N=%d, T threads on one handle"
                                             % (a.limit, a.dim, a.rows),
                                 "runs": [concurrent_callers(a, L, nifs, main_ref, t, 1.0, quantized=100) for t in (1, 8, 64)]}
    # ... and funnel_search callers: up to eight share the sweep of the prefixes
    side["callers_funnel"] = {"workload": "funnel_search stages=[%d] candidates=100 limit=%d, d=%d, N=%d, T threads on one handle"
                                          % (min(a.dim, 128), a.limit, a.dim, a.rows),
                              "runs": [concurrent_callers(a, L, nifs, main_ref, t, 1.0, funnel=min(a.dim, 128)) for t in (1, 8, 64)]}
    # config 2: flat cosine top-10, N = 1M, single query
    rows2 = min(1_000_000, a.rows)
    x = build_shard(torch, device, rows2, a.dim, SEED_CORPUS + 2)
    ref2 = nifs.flat_new_cosine()
    nifs.flat_set_reduce_order(ref2, ORDER_CODE[a.reduce_order])
    assert nifs.flat_load_device_matrix(ref2, doc_ids(0, rows2), x.data_ptr(), rows2, a.dim) == ("ok", ())
    del x
    qs = normalized_queries(1100, a.dim, SEED_QUERY + 2)
    side["config2"] = dict(leg(a, L, nifs, ref2, "single", qs, 1000, 100),
                           workload="index: :flat, metric: :cosine, d=%d, N=%d, single query" % (a.dim, rows2), dtype="f32")
    del ref2
    # config 3: dot, N rows, batches of 256 (rows scaled x U(8,24): no normalisation, collection.ex:1302, :1319)
    x = build_shard(torch, device, a.rows, a.dim, SEED_CORPUS + 3)
    x.mul_(torch.empty((a.rows, 1), device=device).uniform_(8.0, 24.0))
    ref3 = nifs.flat_new_inner_product()
    nifs.flat_set_reduce_order(ref3, ORDER_CODE[a.reduce_order])
    assert nifs.flat_load_device_matrix(ref3, doc_ids(0, a.rows), x.data_ptr(), a.rows, a.dim) == ("ok", ())
    # (the same rows as an L2 collection: the funnel legs below)
    ref0 = nifs._flat_new(0)
    nifs.flat_set_reduce_order(ref0, ORDER_CODE[a.reduce_order])
    assert nifs.flat_load_device_matrix(ref0, doc_ids(0, a.rows), x.data_ptr(), a.rows, a.dim) == ("ok", ())
    del x
    torch.cuda.empty_cache()
    release_pause(a)
    # funnel_search on an L2 collection: stage 1 is K1's arithmetic on the prefix (one caller: K1 itself over the
    # rows' first 128 floats; callers that meet: K1p, up to eight per sweep of the prefixes)
    p128 = min(a.dim, 128)
    qs = np.random.default_rng(SEED_QUERY + 12).uniform(-1, 1, size=(110, a.dim)).astype(np.float32)
    side["funnel_l2"] = dict(leg(a, L, nifs, ref0, "funnel", qs, 100, 10, stages=(p128,), candidates=100, kernel="scan_topk_kernel (prefix)"),
                             workload="index: :flat, metric: :l2, funnel_search stages=[%d] candidates=100 limit=10, d=%d, N=%d" % (p128, a.dim, a.rows),
                             dtype="f32")
    outs64 = (C.c_void_p * 64)()
    qb = np.ascontiguousarray(qs[:64])
    st_arr = (C.c_size_t * 1)(p128)
    nifs.flat_set_profiling(ref0, True)
    times = []
    for rep in range(4):
        nifs.flat_get_profile(ref0, reset=True)
        t1 = time.perf_counter()
        assert L.vt_flat_funnel_search_batch(ref0.handle, qb.ctypes.data_as(C.POINTER(C.c_float)), 64, a.dim, st_arr, 1, 100, a.limit, outs64) == 0
        times.append(time.perf_counter() - t1)
        got = [hits_of(L, C.c_void_p(outs64[j])) for j in range(64)]
    pf = nifs.flat_get_profile(ref0, reset=True)
    nifs.flat_set_profiling(ref0, False)
    for j in range(0, 64, 8):
        h = C.c_void_p()
        assert L.vt_flat_funnel_search(ref0.handle, qb[j].ctypes.data_as(C.POINTER(C.c_float)), a.dim, st_arr, 1, 100, a.limit, C.byref(h)) == 0
        assert hits_of(L, h) == got[j], "batched funnel search differs from the single call"
    sweep_ms = pf["prefix_ms"] / max(1, pf["prefix_launches"])
    side["funnel_l2"]["batch64"] = {
        "ms": min(times[1:]) * 1e3, "queries_per_s": 64 / min(times[1:]), "sweeps": pf["prefix_launches"],
        "queries_in_sweeps": pf["prefix_queries"], "kernel": "prefix_multi_kernel", "avg_sweep_ms": sweep_ms,
        "sweep_GBps": pf["prefix_bytes"] / max(1, pf["prefix_launches"]) / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0,
        "verified": True}
    side["callers_funnel_l2"] = {"workload": "index: :flat, metric: :l2, funnel_search stages=[%d] candidates=100 limit=%d, d=%d, N=%d, "
                                             "T threads on one handle" % (p128, a.limit, a.dim, a.rows),
                                 "runs": [concurrent_callers(a, L, nifs, ref0, t, 1.0, funnel=p128) for t in (1, 8, 64)]}
    del ref0
    torch.cuda.empty_cache()
    qs = np.random.default_rng(SEED_QUERY + 3).uniform(-1, 1, size=(7 * 256, a.dim)).astype(np.float32)
    # ... candidates nominated on the FP32 matrix cores (K2, the r01/r02 path, unchanged) ...
    assert nifs.flat_set_batch_nominate(ref3, _lib_consts().NOMINATE_F32) == "ok"
    side["config3"] = dict(leg(a, L, nifs, ref3, "batch", qs, 5, 2, per=256),
                           workload="index: :flat, metric: :dot, d=%d, N=%d, batch=256 queries (FP32 MFMA Q x D^T + exact rescoring)"
                           % (a.dim, a.rows), dtype="f32")
    # ... and on the bf16 matrix cores (K2b, the library's default since r03): same hits bit for bit
    assert nifs.flat_set_batch_nominate(ref3, _lib_consts().NOMINATE_BF16) == "ok"
    qs = np.random.default_rng(SEED_QUERY + 4).uniform(-1, 1, size=(24 * 256, a.dim)).astype(np.float32)
    side["config3_bf16_nominate"] = dict(
        leg(a, L, nifs, ref3, "batch", qs, 20, 4, per=256),
        workload="index: :flat, metric: :dot, d=%d, N=%d, batch=256 queries (bf16 MFMA nomination, HBM-bound, + exact f32 rescoring)"
        % (a.dim, a.rows), dtype="f32 (exact rescoring; bf16 nomination)")
    # ... and the config as BASELINE.json writes it -- "16 batches x 256"  -- as ONE vt_flat_search_batch of
    # 4 096 queries: consecutive groups of 256 alternate between two contexts, group g + 1's upload / sample / threshold
    # beside group g's exact rescoring, group g's verdicts and hit lists under group g + 1's pass over the rows
    qs = np.random.default_rng(SEED_QUERY + 5).uniform(-1, 1, size=(5 * 4096, a.dim)).astype(np.float32)
    r16 = leg(a, L, nifs, ref3, "batch", qs, 4, 1, per=4096)
    side["config3_bf16_nominate"]["one_call_16x256"] = {
        k: r16[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "fallback_queries", "second_passes",
                            "candidates_per_query", "end_to_end_frac", "passes_per_step", "verified")}
    side["config3_bf16_nominate"]["one_call_16x256"]["kernel_avg_launch_ms"] = r16["roofline"]["avg_launch_ms"]
    del ref3
    torch.cuda.empty_cache()
    # float hamming (distances.rs:319-324) on N sparse rows: flat_search reads the non-zero-bit column
    # (K4) -- N * ceil(d / 64) * 8 algorithmic bytes per search instead of N * d * 4
    x = build_shard(torch, device, a.rows, a.dim, SEED_CORPUS + 7)
    g = torch.Generator(device=device)
    g.manual_seed(SEED_CORPUS + 8)
    for s0 in range(0, a.rows, 1 << 20):
        e0 = min(a.rows, s0 + (1 << 20))
        x[s0:e0] *= (torch.rand((e0 - s0, a.dim), generator=g, device=device) < 0.5)
    ref7 = nifs._flat_new(7)
    assert nifs.flat_load_device_matrix(ref7, doc_ids(0, a.rows), x.data_ptr(), a.rows, a.dim) == ("ok", ())
    del x
    torch.cuda.empty_cache()
    release_pause(a)
    rng = np.random.default_rng(SEED_QUERY + 7)
    qs = (rng.uniform(-1, 1, (330, a.dim)) * (rng.uniform(0, 1, (330, a.dim)) < 0.5)).astype(np.float32)
    side["pattern_hamming"] = dict(leg(a, L, nifs, ref7, "pattern", qs, 300, 30),
                                   workload="index: :flat, metric: :hamming (float), d=%d, N=%d, single query; K4 over the non-zero-bit column"
                                   % (a.dim, a.rows), dtype="u64")
    # ... and a batch of 64 (K4p: eight queries per sweep of the column), every 8th answer against its single search
    outs = (C.c_void_p * 64)()
    qb = np.ascontiguousarray(qs[:64])
    qbp = qb.ctypes.data_as(C.POINTER(C.c_float))
    times = []
    for rep in range(4):
        t1 = time.perf_counter()
        assert L.vt_flat_search_batch(ref7.handle, qbp, 64, a.dim, a.limit, outs) == 0
        times.append(time.perf_counter() - t1)
        got = [hits_of(L, C.c_void_p(outs[j])) for j in range(64)]
    for j in range(0, 64, 8):
        h = C.c_void_p()
        assert L.vt_flat_search(ref7.handle, qb[j].ctypes.data_as(C.POINTER(C.c_float)), a.dim, a.limit, C.byref(h)) == 0
        assert hits_of(L, h) == got[j], "batched pattern search differs from the single search"
    side["pattern_hamming"]["batch64_ms"] = min(times[1:]) * 1e3
    side["pattern_hamming"]["batch64_queries_per_s"] = 64 / min(times[1:])
    # ... and callers that meet on the handle: they travel as such batches (native threads; every 8th answer checked)
    runs = [native_callers(a, L, nifs, ref7, t, 1.0, qb, 0, 0) for t in (8, 64)]
    if all(r is not None for r in runs):
        side["pattern_hamming"]["callers"] = runs
    del ref7
    torch.cuda.empty_cache()
    side["seconds"] = round(time.perf_counter() - t0, 1)
    return side


def batch_roofline(p, rows, dim, dt_per_step=None):
    """The roofline object of a matrix-core batch leg from the library's HIP-event profile `p`:
    K2s / K2b (bf16 nomination: one pass reads every row once for all <= 256 queries -- HBM-bound) or
    K2 (FP32 matrix cores -- MFMA-bound).  Hits are exact either way."""
    if p["nominate_launches"]:
        launches = p["nominate_launches"]
        ms = p["nominate_ms"] / launches
        gbs = p["nominate_bytes"] / launches / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        from_shadow = p["nominate_shadow_launches"] == launches
        kern = "shadow_scores_kernel" if from_shadow else "bf16_scores_kernel"
        r = {"bound": "hbm", "kernel": kern, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": gbs / HBM_PEAK_GBS, "traffic": pmc_side_traffic(kern, rows, dim),
             "traffic_source": PMC_SIDE_SOURCE if pmc_side_traffic(kern, rows, dim) is not None else None,
             "avg_launch_ms": ms, "algorithmic_bytes_per_launch": p["nominate_bytes"] / launches,
             "bound_note": ("at 256 columns neither roof: the part holds 1.72-1.88 GHz under bf16 MFMA at > 1 PFLOP/s beside 4 TB/s of "
                            "reads (2.38 GHz without the row DMA; profiles/r04_k2s_pmc.txt, DESIGN 4.5); 128 / 64 columns: 0.81 / 0.91 of HBM"
                            if from_shadow else None),
             "algorithmic_bytes_note": ("rows x d x 2: the pass reads the bf16 shadow of the rows (K2s)" if from_shadow else
                                        "rows x d x 4: the pass streams the f32 rows and rounds them in registers (K2b)")
                                       + "; the exact rescoring of ~%d gathered rows per query is a launch of its own"
                                       % round(p["nominate_candidates"] / max(1, p["nominate_queries"])),
             "bf16_mfma_TFLOPs": p["nominate_flops"] / launches / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
             "bf16_mfma_frac_of_2500_TFLOPs": p["nominate_flops"] / launches / (ms * 1e-3) / 1e12 / 2500.0 if ms > 0 else 0.0}
        return r
    ms = p["batch_ms"] / max(1, p["batch_launches"])
    tf = p["batch_flops"] / max(1e-9, p["batch_ms"]) / 1e9
    return {"bound": "mfma", "kernel": "mfma_scores_kernel", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
            "traffic": pmc_side_traffic("mfma_scores_kernel", rows, dim),
            "traffic_source": PMC_SIDE_SOURCE if pmc_side_traffic("mfma_scores_kernel", rows, dim) is not None else None,
            "avg_launch_ms": ms, "algorithmic_flops_per_launch": p["batch_flops"] / max(1, p["batch_launches"])}


def measure_batches(a, torch, dist, nifs, _lib, L, ref, sharded, use_dist, launched, rank, world, devices, shards_in_process,
                    force_sharded, total_rows, count, rccl_ranks, sharding, host_exchange, device):
    """`--mode batch` on N > 1 GPUs (BASELINE configs[3]'s batched leg: 16 batches of 256 on the
    row-sharded corpus): every shard answers the whole batch on its rows (K2s / K2b / K2 + exact
    rescoring), the per-query lists meet once per batch -- inside the library on a one-process handle
    (host-mapped lists, merge by (rank key, id bytes)), or as ONE all_gather of wire blocks between
    ranks (vettore_amd/sharded.py search_batch) -- and are merged per query."""
    per = a.batch
    assert nifs.flat_set_batch_nominate(ref, _lib.NOMINATE_BF16 if a.nominate == "bf16" else _lib.NOMINATE_F32) == "ok"
    nq = (a.steps + a.warmup) * per
    qs = np.random.default_rng(SEED_QUERY).uniform(-1, 1, size=(nq, a.dim)).astype(np.float32)
    if a.metric == "cosine":
        qs /= np.linalg.norm(qs.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    outs = (C.c_void_p * per)()
    between_ranks = sharded is not None and use_dist

    def step(i, keep=False):
        q = np.ascontiguousarray(qs[i * per:(i + 1) * per])
        if between_ranks:
            res = sharded.search_batch(q, a.limit)
            return [[(h[0], np.float32(h[1]).tobytes()) for h in hits] for hits in res] if keep else None
        st = L.vt_flat_search_batch(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), per, a.dim, a.limit, outs)
        if st != 0:
            sys.exit("bench.py: flat_search_batch failed with status %d: %s" % (st, (L.vt_last_error() or b"").decode()))
        if keep:
            return [hits_of(L, C.c_void_p(outs[j])) for j in range(per)]
        L.vt_hits_free_many(outs, per)
        return None

    def single(q):
        if between_ranks:
            return [(h[0], np.float32(h[1]).tobytes()) for h in sharded.search(q, a.limit)]
        h = C.c_void_p()
        assert L.vt_flat_search(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), a.dim, a.limit, C.byref(h)) == 0
        return hits_of(L, h)

    t_build = time.perf_counter()
    step(0)   # settles id ranks, row norms and the bf16 shadow: setup, not a step
    t_build = time.perf_counter() - t_build
    if a.release_wait > 0:   # (the build's source tensor has just gone back to the driver: see measure())
        time.sleep(a.release_wait)

    def sync():
        if launched:
            dist.barrier()
        for dv in (set(devices) if (shards_in_process > 1 or force_sharded) else ()):
            torch.cuda.synchronize(dv)
        torch.cuda.synchronize()

    def timed_run(profiling):
        for i in range(a.warmup):
            step(i)
        nifs.flat_set_profiling(ref, profiling)
        nifs.flat_get_profile(ref, reset=True)
        sync()
        t0 = time.perf_counter()
        for i in range(a.warmup, a.warmup + a.steps):
            step(i)
        sync()
        dt = time.perf_counter() - t0
        prof = nifs.flat_get_profile(ref, reset=True)
        nifs.flat_set_profiling(ref, False)
        if launched:
            t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cpu") if host_exchange else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, prof

    dt_events, prof = timed_run(True)
    dt, _ = timed_run(False)

    # outside the timed region: the batch equals its queries' single searches (every rank takes part)
    last = step(a.warmup + a.steps - 1, keep=True)
    for j in (0, per - 1):
        assert single(qs[(a.warmup + a.steps - 1) * per + j]) == last[j], "batched result differs from the single-query path"
    if use_dist:
        dist.destroy_process_group()
    if rank != 0:
        return
    out = {
        "metric": "queries/sec, flat %s top-%d, N=%d d=%d, batch=%d on %d GPUs" % (a.metric, a.limit, total_rows, a.dim, per, a.gpus),
        "value": a.steps * per / dt, "unit": "queries/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "ms_per_step_with_event_timing": dt_events / a.steps * 1e3,
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
        "dtype": "f32" if a.nominate == "f32" else "f32 (exact rescoring; bf16 nomination)", "data": "synthetic",
        "rccl_ranks": rccl_ranks,
        "config": {"workload": "index: :flat, metric: :%s, d=%d, N=%d, batch=%d queries (%s MFMA Q x D^T + exact rescoring), rows sharded over %d GPUs"
                               % (a.metric, a.dim, total_rows, per, a.nominate, a.gpus),
                   "rows_per_gpu": count, "reduce_order": a.reduce_order, "sharding": sharding, "processes": world,
                   "setup_s": round(t_build, 1), "fallback_queries": prof["batch_fallbacks"], "verified": True},
        "roofline": batch_roofline(prof, count, a.dim),
    }
    launches = max(1, prof["nominate_launches"] + prof["batch_launches"])
    pass_ms = (prof["nominate_ms"] + prof["batch_ms"]) / launches
    # passes per step and shard: one per 256 queries of the batch (a 4 096-query call pipelines 16 of them)
    groups = launches / a.steps / max(1, shards_in_process)
    out["config"]["per_shard_pass_ms"] = pass_ms
    out["config"]["passes_per_step"] = groups
    # rescoring, select, hand-off, exchange, merge: everything but the passes (what of it hides under the next group's pass
    # inside one call is not in this figure any more)
    out["config"]["exchange_ms"] = dt / a.steps * 1e3 - pass_ms * groups
    if shards_in_process > 1 or force_sharded:
        out["config"]["exchange_note"] = nifs.flat_exchange_note(ref)
        out["config"]["devices"] = devices
    if a.exchange_note:
        out["config"]["exchange_note"] = (out["config"].get("exchange_note", "") + "; " if out["config"].get("exchange_note") else "") + a.exchange_note
    C.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


def config4_leg(a, torch, dist, nifs, _lib, L, rank, world, launched, devices, device, host_exchange):
    """BASELINE configs[3] -- `index: :flat, metric: :l2, d=768, N=40M row-sharded across 8xMI355X, RCCL top-k merge` --
    measured beside the headline when (and only when) the run is `--gpus 8` (the driver passes no flags, so
    the one 8-GPU run there may ever be has to carry this itself): L2, --config4-rows (5 M) rows per GPU generated on the
    device, 200 single queries and ONE call of 4 096 queries, every rank taking part; a roofline of its own (one shard's
    scan: rows x d x 4 bytes = 15.36 GB per GPU and query) and two checks -- a batched list equals its query's single search,
    and a brute-force pass in torch over every shard's own rows finds nothing closer than the reported hits.  Budget: well
    under a minute at the default size between ranks (the in-process form, `--devices`, builds every shard in turn and is
    for rehearsals with a small --config4-rows).  Returns the leg's dictionary (identical on every rank)."""
    from vettore_amd.sharded import ShardedFlat
    t_leg = time.perf_counter()
    if launched and host_exchange and not a.config4_anyway:
        # (the fallback child between ranks: gloo, no RCCL -- nothing of configs[3]'s "RCCL top-k merge" would be measured)
        return {"skipped": "this child runs over the host exchange between ranks (the fallback): the config 4 leg is measured over RCCL only"}
    xdev = torch.device("cpu") if host_exchange else device   # where collectives' tensors live (gloo: the host)
    rows, dim, limit = a.config4_rows, a.dim, a.limit
    metric = nifs.METRIC_CODE["l2"]
    between_ranks = launched or a.force_exchange   # (--force-exchange: the rank-per-GPU form with the one rank of a one-GPU box)
    n_gpus = world if between_ranks else len(devices)
    total = rows * n_gpus
    nsingle, nbatch = 200, 4096
    rng = np.random.default_rng(SEED_QUERY + 4)
    qs = rng.uniform(-1, 1, size=(nsingle, dim)).astype(np.float32)
    qb = rng.uniform(-1, 1, size=(nbatch, dim)).astype(np.float32)
    if between_ranks:
        x = build_shard(torch, device, rows, dim, SEED_CORPUS + 400 + rank, normalize=False)
        ids = doc_ids(rank * rows, rows)
        ref = nifs._flat_new(metric)
        nifs.flat_set_reduce_order(ref, ORDER_CODE[a.reduce_order])
        res = nifs.flat_load_device_matrix(ref, ids, x.data_ptr(), rows, dim)
        assert res == ("ok", ()), res
        # (64-byte records over RCCL, merged on the host: no global id ranking to pay for)
        sf = ShardedFlat(ref, dist, xdev, force_exchange=a.force_exchange)
        rccl_ranks = 0 if host_exchange else dist.get_world_size()
        exchange = "one rank per GPU: all_gather of per-shard top-k records over %s, merge by (rank key, id bytes) on the host" % (
            "gloo" if host_exchange else "RCCL")

        def single(q):
            return [(h[0], np.float32(h[1]).tobytes()) for h in sf.search(q, limit)]

        def batch(Q):
            return [[(h[0], np.float32(h[1]).tobytes()) for h in hits] for hits in sf.search_batch(Q, limit)]
        mine = [(device, x, np.arange(rank * rows + 1, (rank + 1) * rows + 1, dtype=np.int64))]
    else:
        ref = nifs.flat_new_sharded(metric, devices)
        nifs.flat_set_reduce_order(ref, ORDER_CODE[a.reduce_order])
        all_idx = np.arange(1, total + 1, dtype=np.int64)
        route = nifs.flat_route_ids(ref, doc_ids(0, total))
        mine = []
        for s in range(n_gpus):
            idx = all_idx[route == s]
            dev_s = torch.device("cuda", devices[s])
            with torch.cuda.device(dev_s):
                x = build_shard(torch, dev_s, len(idx), dim, SEED_CORPUS + 400 + s, normalize=False)
                res = nifs.flat_load_device_matrix(ref, doc_ids(0, 0, idx), x.data_ptr(), len(idx), dim)
                assert res == ("ok", ()), res
                mine.append((dev_s, None, idx))   # (rebuilt from its seed for the check: eight matrices need not stay)
                del x
                torch.cuda.empty_cache()
        rccl_ranks = nifs.flat_rccl_ranks(ref)
        exchange = "one handle over %d devices (hash of id), %s exchange, merge by (rank key, id bytes)" % (
            n_gpus, "rccl" if nifs.flat_exchange(ref) == _lib.EXCHANGE_RCCL else "host")
        hp = C.c_void_p()
        outs = (C.c_void_p * nbatch)()

        def single(q):
            st = L.vt_flat_search(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), dim, limit, C.byref(hp))
            if st != 0:
                raise RuntimeError("flat_search failed with status %d: %s" % (st, (L.vt_last_error() or b"").decode()))
            return hits_of(L, hp)

        def batch(Q):
            Q = np.ascontiguousarray(Q)
            st = L.vt_flat_search_batch(ref.handle, Q.ctypes.data_as(C.POINTER(C.c_float)), len(Q), dim, limit, outs)
            if st != 0:
                raise RuntimeError("flat_search_batch failed with status %d: %s" % (st, (L.vt_last_error() or b"").decode()))
            return [hits_of(L, C.c_void_p(outs[j])) for j in range(len(Q))]

    def sync():
        if launched:
            dist.barrier()
        for dv in set(d for d, _, _ in mine):
            torch.cuda.synchronize(dv)

    def reduced_max(dt):
        if not launched:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for q in qs[:5]:
        single(q)            # (also settles the id ranks: setup)
    nifs.flat_set_profiling(ref, True)
    nifs.flat_get_profile(ref, reset=True)
    sync()
    t0 = time.perf_counter()
    for q in qs:
        hits = single(q)
    sync()
    dt_single = reduced_max(time.perf_counter() - t0)
    prof = nifs.flat_get_profile(ref, reset=True)
    nifs.flat_set_profiling(ref, False)
    batch(qb[:256])          # (norms, the bf16 shadow: setup)
    sync()
    t0 = time.perf_counter()
    lists = batch(qb)
    sync()
    dt_batch = reduced_max(time.perf_counter() - t0)
    # ---- checks, outside every timed region ------------------------------------------------------------------------------
    batched_equals_single = all(lists[j] == single(qb[j]) for j in (0, nbatch // 2 - 1, nbatch - 1))
    # brute force over every shard's own rows: nothing closer than the k-th reported hit has been missed, and the reported
    # hits' distances are the rows' (f32 arithmetic in another order: 1e-5 relative, north_star's tolerance)
    check_ok, better = True, 0
    for q in (qs[0], qs[nsingle - 1]):
        hits = single(q)
        raw = {h[0]: float(np.frombuffer(h[1], dtype=np.float32)[0]) for h in hits}
        kth = max(raw.values())
        for dv, x, idx in mine:
            with torch.cuda.device(dv):
                if x is None:
                    s_no = [i for i, m in enumerate(mine) if m[2] is idx][0]
                    xs = build_shard(torch, dv, len(idx), dim, SEED_CORPUS + 400 + s_no, normalize=False)
                else:
                    xs = x
                qv = torch.from_numpy(q).to(dv)
                dist2 = torch.empty(len(idx), dtype=torch.float64, device=dv)
                for lo in range(0, len(idx), 1 << 20):
                    hi = min(len(idx), lo + (1 << 20))
                    dist2[lo:hi] = (xs[lo:hi].double() - qv.double()).pow(2).sum(dim=1)
                d = dist2.sqrt()
                better += int((d < kth * (1.0 - 1e-5)).sum().item())
                for key, r in raw.items():
                    num = int(key[4:])                                # ids are "doc-<number>", idx is ascending
                    pos = int(np.searchsorted(idx, num))
                    if pos >= len(idx) or int(idx[pos]) != num:
                        continue                                      # (another shard's row)
                    ref_d = float(d[pos].item())
                    if abs(r - ref_d) > 1e-5 * max(1.0, abs(ref_d)):
                        check_ok = False
                if x is None:
                    del xs
                    torch.cuda.empty_cache()
    if launched:
        t = torch.tensor([better, 0 if check_ok else 1], dtype=torch.int64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        better, check_ok = int(t[0].item()), int(t[1].item()) == 0
    # (two queries, ten hits each: at most nine rows are strictly closer than a query's tenth hit)
    brute_force_ok = check_ok and better <= 2 * (limit - 1)
    launches = max(1, prof["scan_launches"])
    scan_ms = prof["scan_ms"] / launches
    bytes_per_launch = prof["scan_bytes"] / launches
    achieved = bytes_per_launch / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    del mine
    return {
        "workload": "index: :flat, metric: :l2, d=%d, N=%d row-sharded across %d GPUs (%d rows each), RCCL top-k merge; "
                    "%d single queries, then %d queries in one call" % (dim, total, n_gpus, rows, nsingle, nbatch),
        "n_gpus": n_gpus, "rows_per_gpu": rows, "rccl_ranks": rccl_ranks, "exchange": exchange, "dtype": "f32", "data": "synthetic",
        "single": {"value": nsingle / dt_single, "unit": "queries/s", "steps": nsingle, "ms_per_step": dt_single / nsingle * 1e3,
                   "roofline": {"bound": "hbm", "kernel": "scan_topk_kernel (one shard's scan)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": scan_ms},
                   "everything_but_the_scan_ms": dt_single / nsingle * 1e3 - scan_ms},
        "batch_4096_one_call": {"value": nbatch / dt_batch, "unit": "queries/s", "ms": dt_batch * 1e3},
        "verified": bool(batched_equals_single and brute_force_ok),
        "checks": {"batched_list_equals_single_search": bool(batched_equals_single), "brute_force_over_every_shard": bool(brute_force_ok),
                   "rows_closer_than_a_tenth_hit": better},
        "seconds": round(time.perf_counter() - t_leg, 1),
    }


def json_line_of(text):
    """The last line of `text` that parses as a JSON object (a child's result), or None."""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def run_child(argv, exchange, note, port_shift, timeout):
    """One supervised measurement: this script again with --child, in a process (group) of its own.
    Returns (exit status or -9 after a timeout, its stdout, seconds).  The supervisor itself never
    touches a GPU: a process that has initialised one cannot be replaced by another program, and a
    wedged collective can only be left behind by ending the process that holds it."""
    import signal
    import subprocess
    # (VT_BENCH_CHILD: tests/test_bench_supervisor.py puts a stand-in for the measurement there -- the
    # supervisor's logic runs on a CPU box)
    script = os.environ.get("VT_BENCH_CHILD") or os.path.abspath(__file__)
    cmd = [sys.executable, script] + [v for v in argv if v != "--supervise"] + ["--child", "--exchange", exchange]
    if note:
        cmd += ["--exchange-note", note]
    env = dict(os.environ)
    if port_shift and "MASTER_PORT" in env:
        # a rendezvous the first run left half open stays out of the way: the second run's ranks meet on the next port --
        # where rank 0 must host the store ITSELF: under torch.distributed.run the launcher's agent hosts the one at
        # MASTER_PORT and tells its workers to come as clients only (TORCHELASTIC_USE_AGENT_STORE); nobody listens one
        # port up (found with the driver's own command on one GPU: both ranks' clients timed out after 2 x 120 s)
        env["MASTER_PORT"] = str(int(env["MASTER_PORT"]) + port_shift)
        env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
    t0 = time.perf_counter()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)   # the group this call started, nothing else
        except ProcessLookupError:
            pass
        out, _ = proc.communicate()
        rc = -9
    return rc, out.decode(errors="replace"), time.perf_counter() - t0


def launch_key():
    """What names THIS launch and nothing else: the launcher's pid together with its start time (a recycled pid has
    another one), torch's run id and the rendezvous port.  Every rank of a launch computes the same key."""
    ppid = os.getppid()
    started = "0"
    try:
        with open("/proc/%d/stat" % ppid) as f:
            started = f.read().rsplit(")", 1)[1].split()[19]   # field 22: start time in clock ticks since boot
    except (OSError, IndexError):
        pass
    run_id = "".join(ch for ch in os.environ.get("TORCHELASTIC_RUN_ID", "none") if ch.isalnum())[:32]
    return "%d_%s_%s_%s" % (ppid, started, run_id, os.environ.get("MASTER_PORT", "0"))


def ranks_agree(tag, rank, world, ok, wait_s):
    """Under torch.distributed.run every rank supervises its own child, and all of them must take
    the same next step (a rank that goes on to the host exchange alone would wait for the others
    for ever).  Each writes its verdict into a directory named after this launch (launch_key: a directory left behind
    by an earlier launch, or made by somebody else, is never this one -- ADVICE r4) and reads the others'; a rank that
    never reports counts as failed.  Rank 0 removes the directory once every rank has read it."""
    import shutil
    import tempfile
    d = os.path.join(tempfile.gettempdir(), "vt_bench_%s_%s" % (launch_key(), tag))
    os.makedirs(d, mode=0o700, exist_ok=True)
    if os.stat(d).st_uid != os.getuid():
        return False   # (somebody else's directory under our name: no agreement through it)
    tmp = os.path.join(d, "rank%d.tmp" % rank)
    with open(tmp, "w") as f:
        f.write("1" if ok else "0")
    os.replace(tmp, os.path.join(d, "rank%d" % rank))
    deadline = time.perf_counter() + wait_s
    verdict = False
    while time.perf_counter() < deadline:
        seen = []
        for r in range(world):
            try:
                with open(os.path.join(d, "rank%d" % r)) as f:
                    seen.append(f.read().strip())
            except OSError:
                break
        if len(seen) == world:
            verdict = all(v == "1" for v in seen)
            break
        time.sleep(0.05)
    else:
        return False
    try:
        open(os.path.join(d, "read%d" % rank), "w").close()
        if rank == 0:
            until = time.perf_counter() + 5.0
            while time.perf_counter() < until and not all(os.path.exists(os.path.join(d, "read%d" % r)) for r in range(world)):
                time.sleep(0.02)
            shutil.rmtree(d, ignore_errors=True)
    except OSError:
        pass
    return verdict


def supervise(a, argv):
    """`bench.py --gpus N` for N > 1 (and `--supervise` at N = 1): the measurement runs in a fresh child
    process over RCCL; should that child exit non-zero or not come back within --child-timeout -- a
    communicator that does not initialise, an all-gather whose peers never arrive (the library then
    fails the search with "RCCL exchange timed out ...", vt_multi.h) -- a second fresh child runs the
    same steps over the host exchange, and its line is printed with config.exchange_note saying why.
    The first multi-GPU run of a new node cannot come back empty."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "WORLD_SIZE" in os.environ and world > 1
    first = a.exchange if a.exchange != "auto" else "rccl"
    attempts = [first] + (["host"] if first != "host" else [])
    why = None
    for i, exchange in enumerate(attempts):
        rc, out, secs = run_child(argv, exchange, why, i, a.child_timeout)
        line = json_line_of(out)
        ok = rc == 0 and (line is not None or rank != 0)
        mine = ok
        if launched:
            ok = ranks_agree("try%d" % i, rank, world, ok, a.child_timeout + 60.0)
        if ok:
            if rank == 0:
                print(json.dumps(line), flush=True)
            return 0
        how = "was killed after %.0f s without an answer" % secs if rc == -9 else \
              "exited with status %d after %.0f s" % (rc, secs) if rc != 0 else \
              ("printed no result line" if not mine else "succeeded here but failed on another rank")
        why = "the %s-exchange run %s; this line is the run over the host exchange" % (exchange, how)
        sys.stderr.write("bench.py[rank %d]: %s\n" % (rank, why if i + 1 < len(attempts) else "the %s-exchange run %s" % (exchange, how)))
        sys.stderr.flush()
    return 1


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "WORLD_SIZE" in os.environ and world > 1
    if not a.child and (a.gpus > 1 or launched or a.supervise):
        sys.exit(supervise(a, sys.argv[1:]))
    return measure(a)


def measure(a):
    steps_default = a.steps is None
    a.steps = 1000 if a.steps is None else a.steps
    a.warmup = 50 if a.warmup is None else a.warmup
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "WORLD_SIZE" in os.environ and world > 1
    if launched:
        a.gpus = world
    shards_in_process = a.gpus if not launched else 1   # one process, one handle over a.gpus devices
    # `--gpus 1 --exchange rccl|host`: the one shard goes through the multi-shard machinery anyway
    # (worker thread, exchange, host merge) -- prices that machinery on a one-GPU box
    force_sharded = not launched and a.gpus == 1 and a.exchange != "auto"
    # weak scaling: --rows is one GPU's share
    total_rows = a.rows * a.gpus if a.scaling == "weak" else a.rows
    normalize = a.metric == "cosine"   # (collection.ex:1317-1319: only cosine collections normalise what they store)
    host_exchange = a.exchange == "host"

    import torch  # first: its bundled libamdhip64 must be the one the process shares
    import torch.distributed as dist
    from vettore_amd import nifs, _lib
    from vettore_amd.sharded import ShardedFlat
    L = _lib.load()
    for item in a.debug_set:   # (--debug-set name=value: vt_debug_set)
        name, _, value = item.partition("=")
        nifs.debug_set(name, int(value or 1))
    if force_sharded:
        nifs.debug_set("shard_force_workers", 1)

    if launched and a.devices:
        # (under a launcher --devices names every rank's GPU: `--devices 0,0` lets a one-GPU box run two ranks,
        # over the host exchange -- RCCL wants a device per rank)
        local_rank = [int(v) for v in a.devices.split(",")][rank]
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = launched or a.force_exchange
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        import datetime
        # (a collective that does not complete fails the run after two minutes instead of torch's ten;
        # the host exchange needs no RCCL at all: 64-byte records over gloo)
        if host_exchange:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=datetime.timedelta(seconds=120))
    nifs.set_device(local_rank)
    multi = a.gpus > 1 or launched or force_sharded or a.force_exchange   # (--force-exchange: the rank-per-GPU path with the one rank)
    if a.mode != "single" and not (a.mode == "batch" and multi):
        if multi:
            sys.exit("--mode %s is a single-GPU measurement" % a.mode)
        if a.mode == "batch" and steps_default:
            a.steps, a.warmup = 8, 2
        return run_side_mode(a, torch, nifs, device)
    if a.mode == "batch" and steps_default:
        a.steps, a.warmup = 16, 2   # BASELINE configs[3]: 16 batches of 256

    t_build = time.perf_counter()
    rccl_ranks = 0
    devices = [local_rank]
    if shards_in_process > 1 or force_sharded:
        # ---- one process, one handle, N devices: rows go where the hash of their id says -------
        ndev = torch.cuda.device_count()
        devices = [int(v) for v in a.devices.split(",")] if a.devices else list(range(shards_in_process))
        if len(devices) != shards_in_process or max(devices) >= ndev:
            sys.exit("--gpus %d needs %d visible devices (or --devices with %d ordinals < %d)" % (
                shards_in_process, shards_in_process, shards_in_process, ndev))
        if a.exchange != "auto":
            nifs.debug_set("shard_exchange", {"host": 1, "rccl": 2}[a.exchange])
        ref = nifs.flat_new_sharded(nifs.METRIC_CODE[a.metric], devices)
        nifs.flat_set_reduce_order(ref, ORDER_CODE[a.reduce_order])
        all_idx = np.arange(1, total_rows + 1, dtype=np.int64)
        route = nifs.flat_route_ids(ref, doc_ids(0, total_rows))
        for s in range(shards_in_process):
            idx = all_idx[route == s]
            dev_s = torch.device("cuda", devices[s])
            with torch.cuda.device(dev_s):
                x = build_shard(torch, dev_s, len(idx), a.dim, SEED_CORPUS + s, normalize=normalize)
                res = nifs.flat_load_device_matrix(ref, doc_ids(0, 0, idx), x.data_ptr(), len(idx), a.dim)
                assert res == ("ok", ()), res
                del x
                torch.cuda.empty_cache()
        count = max(nifs.flat_shard_lens(ref))
        rccl_ranks = nifs.flat_rccl_ranks(ref)
        exchange = "rccl" if nifs.flat_exchange(ref) == _lib.EXCHANGE_RCCL else "host"
        sharding = "one handle over %d devices (hash of id), %s exchange of per-shard top-k, merge by (rank key, id bytes)" % (
            shards_in_process, exchange)
        sharded = None
    else:
        # ---- this rank's row block of the N-row corpus (one device per process) ----------------
        per = total_rows // world
        start = rank * per
        count = per if rank < world - 1 else total_rows - start
        x = build_shard(torch, device, count, a.dim, SEED_CORPUS + rank, normalize=normalize)
        ids = doc_ids(start, count)
        ref = nifs._flat_new(nifs.METRIC_CODE[a.metric])
        nifs.flat_set_reduce_order(ref, ORDER_CODE[a.reduce_order])
        res = nifs.flat_load_device_matrix(ref, ids, x.data_ptr(), count, a.dim)
        assert res == ("ok", ()), res
        del x
        torch.cuda.empty_cache()
        # (over gloo the records travel as host tensors)
        sharded = ShardedFlat(ref, dist if use_dist else None, torch.device("cpu") if host_exchange else device,
                              force_exchange=a.force_exchange)
        if use_dist and not host_exchange and os.environ.get("VT_HOST_EXCHANGE") is None:
            # one ordering of all ids -> shard keys compare on the device (see vettore_amd/sharded.py)
            sharded.enable_device_exchange(ids, max_limit=max(a.limit, 16))
        if use_dist:
            rccl_ranks = dist.get_world_size()
        sharding = ("row blocks, one rank per GPU, all_gather of per-shard top-k over %s (%s merge)"
                    % ("gloo" if host_exchange else "RCCL", "device" if sharded._dev else "host")) if use_dist else "none"
    for sh in ([ref] if a.shadow == "off" else []):
        assert nifs.flat_set_batch_shadow(sh, _lib.SHADOW_OFF) == ("ok", ())

    if a.mode == "batch":
        return measure_batches(a, torch, dist, nifs, _lib, L, ref, sharded, use_dist, launched, rank, world, devices,
                               shards_in_process, force_sharded, total_rows, count, rccl_ranks, sharding, host_exchange, device)

    nq = a.steps + a.warmup
    qs = normalized_queries(nq, a.dim, SEED_QUERY) if normalize else \
        np.random.default_rng(SEED_QUERY).uniform(-1, 1, size=(nq, a.dim)).astype(np.float32)
    hp = C.c_void_p()

    def search_c(q):
        # the step is the C-ABI call itself (the hit list is freed, not unpacked into Python
        # objects -- a NIF would build BEAM terms here)
        st = L.vt_flat_search(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), a.dim, a.limit, C.byref(hp))
        if st != 0:
            # (a wedged exchange arrives here as "RCCL exchange timed out on shard s ...": say so and
            # leave with a non-zero status -- a fresh process is the only retry)
            sys.exit("bench.py: flat_search failed with status %d: %s" % (st, (L.vt_last_error() or b"").decode()))
        n_hits = L.vt_hits_len(hp)
        L.vt_hits_free(hp)
        return range(n_hits)

    search = (lambda q: sharded.search(q, a.limit)) if (sharded is not None and use_dist) else search_c  # noqa: E731
    # first search also settles the id ranks -- setup, not a step
    search(qs[0])
    t_build = time.perf_counter() - t_build
    # Setup's last act (r06): the card is left alone while the driver finishes taking back the build's source tensor.  Found
    # with rocprofv3's kernel trace of this script: for ~1.2 s after `del x; empty_cache()` of the 30-GB generator tensor
    # every scan runs 2.5 % slower (4.57 against 4.45 ms) -- not with the tensor kept, not after a pause, not again after
    # idle stretches of up to 10 s (tools/ramp_probe.py) -- and the driver's --warmup 5 --steps 20 are 0.11 s.
    if a.release_wait > 0:
        time.sleep(a.release_wait)

    def sync():
        if launched:
            dist.barrier()
        for dv in (set(devices) if (shards_in_process > 1 or force_sharded) else ()):
            torch.cuda.synchronize(dv)
        torch.cuda.synchronize()

    def timed_run(profiling, steps=None):
        steps = a.steps if steps is None else steps
        for i in range(a.warmup):
            search(qs[i])
        nifs.flat_set_profiling(ref, profiling)
        nifs.flat_get_profile(ref, reset=True)
        sync()
        t0 = time.perf_counter()
        for i in range(a.warmup, a.warmup + steps):
            hits = search(qs[i])
        sync()
        dt = time.perf_counter() - t0
        prof = nifs.flat_get_profile(ref, reset=True)
        nifs.flat_set_profiling(ref, False)
        assert len(hits) == min(a.limit, total_rows)
        if launched:
            t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cpu") if host_exchange else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, prof

    # The K timed steps twice, like every side leg: first WITH the library's
    # HIP-event bookkeeping -- the dominant kernel's own duration, measured live on the stream it is
    # launched on: the roofline figure --, then WITHOUT it -- two event records per call are two
    # barrier packets in a chain of three launches -- end to end: `value` and `ms_per_step`.
    #
    # r06 -- the build's last act is to hand its 30-GB source tensor back to the driver, and for ~1.2 s after such a release
    # every scan runs 2.5 % slower (release_wait above): rounds 1-5 timed the driver's 20 steps inside that stretch.
    dt_events, prof = timed_run(True)
    dt, _ = timed_run(False)

    # A timed region of a fraction of a second (the driver's --steps 20 is 0.09 s of scans) is one number without a
    # spread: the same step is then timed once more for >= 1.2 s -- same barrier + synchronize bracket, same maximum
    # over the ranks, the queries taken round and round -- and reported beside it as `long_run`; `steps` / `warmup` /
    # `value` / `ms_per_step` keep the driver's semantics.  (Every rank derives the count from the reduced dt.)
    long_run = None
    if dt < 1.0 and a.steps > 0:
        long_steps = int(math.ceil(1.2 / (dt / a.steps)))
        each = np.empty(long_steps)
        sync()
        t0 = time.perf_counter()
        for i in range(long_steps):
            t_i = time.perf_counter()
            search(qs[i % nq])
            each[i] = time.perf_counter() - t_i
        sync()
        dt_long = time.perf_counter() - t0
        if launched:
            t = torch.tensor([dt_long], dtype=torch.float64, device=torch.device("cpu") if host_exchange else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_long = float(t.item())
        # (r06: every step of it timed as well.  The 255th search call of a fresh process takes ~40 ms -- once, whatever the
        # corpus: a one-time event of the runtime by the look of it, DESIGN 5 -- and with the driver's W and K that call falls
        # into this loop and moves its mean by 3 %: the median and the longest step say so)
        long_run = {"steps": long_steps, "ms_per_step": dt_long / long_steps * 1e3, "value": long_steps / dt_long,
                    "seconds": dt_long, "ms_per_step_median": float(np.median(each)) * 1e3, "longest_step_ms": float(each.max()) * 1e3,
                    "steps_over_twice_the_median": int((each > 2.0 * np.median(each)).sum())}

    out = None
    if rank == 0:
        qps = a.steps / dt
        scan_ms = prof["scan_ms"] / max(1, prof["scan_launches"])
        bytes_per_launch = prof["scan_bytes"] / max(1, prof["scan_launches"])
        achieved = bytes_per_launch / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        out = {
            "metric": "queries/sec, flat %s top-%d, N=%d d=%d (achieved HBM GB/s in roofline)" % (a.metric, a.limit, total_rows, a.dim),
            "value": qps,
            "unit": "queries/s",
            "n_gpus": a.gpus,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "ms_per_step_with_event_timing": dt_events / a.steps * 1e3,
            "long_run": long_run,
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "rccl_ranks": rccl_ranks,
            "config": {
                "workload": "index: :flat, metric: :%s, d=%d, N=%d, limit=%d, single query in flight" % (a.metric, a.dim, total_rows, a.limit),
                "rows_per_gpu": count,
                "reduce_order": a.reduce_order,
                "sharding": sharding,
                "processes": world,
                "setup_s": round(t_build, 1),
                "release_wait_s": a.release_wait,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "scan_topk_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(count, a.dim),
                "traffic_source": PMC_SOURCE if pmc_traffic(count, a.dim) is not None else None,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": scan_ms,
                "measured_read_peak": measured_read_peak(devices[0]),
                "frac_of_measured_read_peak": (achieved / measured_read_peak(devices[0])) if measured_read_peak(devices[0]) else None,
            },
        }
        if shards_in_process > 1 or force_sharded or launched:
            # one line must be enough to read a bad scaling curve: the scan a shard does per query
            # (HIP events, averaged over shards and steps) and everything else of a step -- query
            # upload, select, worker hand-off, the exchange itself, D2H, host merge
            out["config"]["per_shard_scan_ms"] = scan_ms
            out["config"]["exchange_ms"] = dt / a.steps * 1e3 - scan_ms
        if shards_in_process > 1 or force_sharded:
            out["config"]["exchange_note"] = nifs.flat_exchange_note(ref)
            out["config"]["devices"] = devices
        if a.exchange_note:
            out["config"]["exchange_note"] = (out["config"].get("exchange_note", "") + "; " if out["config"].get("exchange_note") else "") + a.exchange_note
            if len(set(devices)) < len(devices):
                # several shards on one GPU: their scans overlap, a launch sees a share of the card
                out["roofline"]["note"] = "shards share a device: per-launch figures are one overlapping scan's share"
                out["roofline"]["whole_job_GBps"] = total_rows * a.dim * 4 / (dt / a.steps) / 1e9 / len(set(devices))
        if (shards_in_process > 1 or force_sharded) and rccl_ranks:
            # the same steps over the other exchange, for comparison (not the headline)
            assert nifs.flat_set_exchange(ref, _lib.EXCHANGE_HOST) == "ok"
            dt2, _ = timed_run(False)
            out["config"]["host_exchange_ms_per_step"] = dt2 / a.steps * 1e3
            assert nifs.flat_set_exchange(ref, _lib.EXCHANGE_RCCL) == "ok"
        if a.gpus == 1 and not launched and not a.no_side and not force_sharded and a.metric == "cosine":
            out["side"] = side_legs(a, torch, nifs, L, device, ref)
        if a.gpus == 1 and not launched and not a.no_cpu and a.cpu_seconds > 0:
            cb = cpu_baseline(a.dim, a.limit, a.cpu_seconds)
            out["cpu_baseline"] = {
                "value": cb["rows_per_s_T"] / a.rows,
                "unit": "queries/s",
                "cores": cb["threads"],
                "kind": "port",
                "sample": "reference-shaped search (flat.rs:96-124), -O3 for baseline x86-64, %d concurrent readers for %.1f s over a "
                          "%d-row x %d sample of the same corpus (%.1f GB: beyond the last-level cache); rows/s scaled to N=%d; "
                          "`variants`: the same sample under both builds, both layouts, 1 and T threads (%.0f s in all)"
                          % (cb["threads"], cb["seconds_per_leg"], cb["sample_rows"], a.dim, cb["sample_bytes"] / 1e9, a.rows, cb["seconds"]),
                "single_thread_value": cb["rows_per_s_1"] / a.rows,
                "effective_GBps": cb["rows_per_s_T"] * a.dim * 4 / 1e9,
                "single_thread_effective_GBps": cb["rows_per_s_1"] * a.dim * 4 / 1e9,
                "variants": [{"build": v["build"], "shape": v["shape"], "threads": v["threads"],
                              "value": v["rows_per_s"] / a.rows, "effective_GBps": v["effective_GBps"]} for v in cb["variants"]],
            }
    # BASELINE configs[3] beside the headline -- at the width of the node, and only there (every rank takes part)
    if (a.gpus == 8 or a.config4_anyway) and not a.no_side:
        # (a failure of the side leg must not take the headline with it: every rank runs the same code on the same seeds,
        # so what raises on one raises on all of them, at the same place)
        try:
            leg4 = config4_leg(a, torch, dist, nifs, _lib, L, rank, world, launched, devices, device, host_exchange)
        except Exception as e:  # noqa: BLE001
            leg4 = {"failed": "%s: %s" % (type(e).__name__, e), "verified": False}
        if rank == 0:
            out.setdefault("side", {})["config4"] = leg4
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # The handful of numbers a reader of a TRUNCATED record needs (VERDICT r5 #4b: the driver keeps `config` whole
        # and the last 2 000 bytes of the line): once inside `config`, once as the line's LAST key.
        side = out.get("side") or {}

        def at(*path):
            o = side
            for k in path:
                o = o.get(k) if isinstance(o, dict) else None
            return o
        summary = {
            "long_run_queries_per_s": long_run["value"] if long_run else None,
            "long_run_median_ms_per_step": long_run["ms_per_step_median"] if long_run else None,
            "kernel_frac_of_8TBps": out["roofline"]["frac"],
            "config2_end_to_end_frac": at("config2", "end_to_end_frac"),
            "config3_16x256_one_call_queries_per_s": at("config3_bf16_nominate", "one_call_16x256", "value"),
            "config3_k2s_frac": at("config3_bf16_nominate", "roofline", "frac"),
            "config5_end_to_end_frac": at("config5", "end_to_end_frac"),
            "config5_kernel_frac": at("config5", "roofline", "frac"),
            "oracle_verified": (all(at("verified_against_oracle", "equal_bit_for_bit").values())
                                if at("verified_against_oracle", "equal_bit_for_bit") else None),
        }
        if at("config4"):
            summary["config4_single_queries_per_s"] = at("config4", "single", "value")
            summary["config4_single_scan_frac"] = at("config4", "single", "roofline", "frac")
            summary["config4_batch_4096_queries_per_s"] = at("config4", "batch_4096_one_call", "value")
            summary["config4_verified"] = at("config4", "verified")
        out["config"]["summary"] = summary
        out["summary"] = summary
        # RCCL's version banner sits in libc's stdout buffer: flush it first so
        # that the JSON line is the last line of output
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
