"""The erl_nif shim EXECUTED against the real library on the GPU: every
NIF of integration/c_src/vettore_gpu_nif.c is called through the ErlNifFunc table with terms
built by the fake runtime (tests/nif_runtime.py), and what comes back -- {:ok, {}}, hit lists,
error strings, ArgumentError -- is compared with the reference's own fixtures
(tests/golden/flat_rs.json, elixir_nif.json), with the oracle bit for bit, and with the Python
mirror of Vettore.Nifs going through the same C ABI.  Only the BEAM itself is still missing.
"""
import struct

import numpy as np
import pytest

import nif_runtime
from nif_runtime import ArgumentError, Atom, FloatList, OK, ERROR
from support import b, load, run_steps

pytestmark = pytest.mark.gpu

UNIT = (OK, ())   # Ok(()) as rustler encodes it (vector_algorithms_hardening_test.exs:56)


@pytest.fixture(scope="module")
def rt():
    import vettore_amd._lib as L
    assert L.load().vt_device_count() >= 1, "no HIP device: GPU tests need the real hardware"
    return nif_runtime.Runtime()


class NifError(Exception):
    pass


def ok(res):
    if res[0] == OK:
        return res[1]
    assert res[0] == ERROR and isinstance(res[1], bytes), res
    raise NifError(res[1].decode())


def bits(hits):
    return [(h[0], np.float32(h[1]).tobytes()) for h in hits]


class NifIndex:
    """support.run_steps adapter: every step goes through the shim AND the oracle's FlatIndex;
    search results must agree bit for bit, errors must be the same strings.  (The reference's
    NIF surface has no len / dimension call: those two are the oracle's.)"""

    def __init__(self, rt, oracle_mod, metric_code):
        self.rt, self.ref = rt, rt.call("flat_new", metric_code, [0])
        assert isinstance(self.ref, nif_runtime.Resource)
        self.o = oracle_mod.FlatIndex(metric_code)
        self.oerr = oracle_mod.OracleError

    def _both(self, nif_call, oracle_call):
        want_err = None
        try:
            want = oracle_call()
        except self.oerr as e:
            want_err = str(e)
        try:
            got = ok(nif_call())
        except NifError as e:
            assert want_err == str(e), (want_err, str(e))
            raise
        assert want_err is None, want_err
        return got, want

    def insert(self, id_, vector):
        got, _ = self._both(lambda: self.rt.call("flat_insert", self.ref, b(id_), [float(x) for x in vector]),
                            lambda: self.o.insert(id_, vector))
        assert got == ()

    def insert_many(self, items):
        got, _ = self._both(lambda: self.rt.call("flat_insert_many", self.ref, [(b(i), [float(x) for x in v]) for i, v in items]),
                            lambda: self.o.insert_many(items))
        assert got == ()

    def delete(self, id_):
        assert self.rt.call("flat_delete", self.ref, b(id_)) == UNIT
        self.o.delete(id_)

    def search(self, query, limit):
        got, want = self._both(lambda: self.rt.call("flat_search", self.ref, [float(x) for x in query], limit),
                               lambda: self.o.search(query, limit))
        assert bits(got) == bits(want)
        return got

    def __len__(self):
        return len(self.o)

    @property
    def dimension(self):
        return self.o.dimension


def _non_finite(step):
    vals = list(step.get("vector", [])) + list(step.get("query", []))
    for it in step.get("items", []):
        vals += list(it[1])
    return any(not np.isfinite(v) for v in vals)


def test_flat_rs_scripts_through_the_shim(rt, oracle_mod):
    """flat.rs:164-303 step scripts (upsert, delete, tie-break by id, validation order, atomic
    batches, limit = usize::MAX, duplicate ids in one batch, f64 recovery) -- each step through
    flat_insert / flat_insert_many / flat_delete / flat_search of the shim."""
    before = rt.live_resources()
    skipped = 0
    for case in load("flat_rs.json"):
        if case.get("differential"):
            continue
        # A BEAM float is always finite: the steps that feed NaN / infinity to FlatIndex directly
        # (flat.rs:193-195, :203-204) have no term to arrive in -- the shim's decoder refuses such
        # a double like rustler's would (test_decode_failures_are_badarg); they run at the C ABI
        # level in tests/test_gpu_parity.py.
        steps = [st for st in case["steps"] if not _non_finite(st)]
        skipped += len(case["steps"]) - len(steps)
        ix = NifIndex(rt, oracle_mod, oracle_mod.METRIC_CODE[case["metric"]])
        run_steps(ix, steps, NifError)
        ix.ref.release()
    assert rt.live_resources() == before and skipped <= 4


def test_flat_rs_differential_all_metrics(rt, oracle_mod):
    case = next(c for c in load("flat_rs.json") if c.get("differential"))
    for name in case["metrics"]:
        ix = NifIndex(rt, oracle_mod, oracle_mod.METRIC_CODE[name])
        ix.insert_many([(r[0], r[1]) for r in case["rows"]])
        for limit in case["limits"]:
            ix.search(case["query"], limit)
        ix.ref.release()


def test_elixir_fixtures_at_the_nif_level(rt, oracle_mod):
    d = load("elixir_nif.json")
    # all nine metrics, rows b=[0,1], a=[1,0], c=[1,0] (cosine collections normalise first: unit rows already)
    c = d["all_supported_metrics_return_stable_top_k_results"]
    for name in c["metrics"]:
        ref = rt.call("flat_new", oracle_mod.METRIC_CODE[name], [0])
        assert rt.call("flat_insert_many", ref, [(b(r[0]), r[1]) for r in c["rows"]]) == UNIT
        hits = ok(rt.call("flat_search", ref, c["query"], c["limit"]))
        assert [h[0] for h in hits] == [b(x) for x in c["expect_ids"]], name
        ref.release()
    # {:ok, {}} and the reference's error strings
    c = d["phantom_native_id_and_ok_unit"]
    ref = rt.call("flat_new", 0, [0])
    assert rt.call("flat_insert", ref, b(c["flat_insert"][0]), c["flat_insert"][1]) == UNIT
    assert rt.call("flat_insert", ref, b"bad", []) == (ERROR, c["put_empty_error"].encode())
    assert rt.call("flat_insert", ref, b"bad", [1.0, 2.0]) == (ERROR, b"dimension mismatch")
    assert rt.call("flat_insert_many", ref, []) == UNIT                          # empty batch (flat.rs:251-253)
    assert rt.call("flat_insert", ref, b"", [5.0]) == UNIT                       # the empty id is an id
    assert ok(rt.call("flat_search", ref, [5.0], 1)) == [(b"", 0.0)]
    assert rt.call("flat_delete", ref, b"never there") == UNIT
    assert ok(rt.call("flat_search", ref, [0.0], 0)) == []
    assert rt.call("flat_search", ref, [0.0, 1.0], 0) == (OK, [])                # limit 0 before validation (flat.rs:97-101)
    assert rt.call("flat_search", ref, [0.0, 1.0], 1) == (ERROR, b"dimension mismatch")
    assert rt.call("flat_search", ref, [], 1) == (ERROR, b"vector must not be empty")
    with pytest.raises(ArgumentError):
        rt.call("flat_search", ref, [0.0], -1)
    with pytest.raises(ArgumentError):
        rt.call("flat_search", ref, [0], 1)
    assert ok(rt.call("flat_search", ref, [0.0], 2 ** 64 - 1)) == [(b"phantom", 0.0), (b"", 5.0)]   # usize::MAX
    ref.release()
    # vector_top_k / binary_top_k: shapes, codes, errors (vector_algorithms_hardening_test.exs:90-106)
    c = d["batched_native_helpers"]
    vecs = [(b(v[0]), v[1]) for v in c["vectors"]]
    for mc in c["metric_codes"]:
        hits = ok(rt.call("vector_top_k", vecs, c["query"], mc, c["dimensions"], c["limit"]))
        assert [h[0] for h in hits] == [b(x) for x in c["expect_ids"]] and all(isinstance(h[1], float) for h in hits), mc
    assert rt.call("vector_top_k", vecs, c["query"], c["unknown_metric"][0], 2, 2) == (ERROR, c["unknown_metric"][1].encode())
    assert rt.call("vector_top_k", vecs, c["query"], 0, 0, 2) == (ERROR, c["bad_prefix"][2].encode())
    assert ok(rt.call("vector_top_k", [], c["query"], 0, 2, 2)) == []
    bn = c["binary"]
    assert ok(rt.call("binary_top_k", [(b(v[0]), v[1]) for v in bn["vectors"]], bn["query"], bn["dimensions"], bn["limit"])) == \
        [(b(e[0]), e[1]) for e in bn["expect"]]
    # compress_sign_bits returns a bare list, normalize_l2 {:ok, list} (nifs.rs:107-129)
    assert rt.call("compress_sign_bits", [1.0, -1.0, 0.0]) == [5]
    assert rt.call("compress_sign_bits", []) == []
    assert rt.call("normalize_l2", [3.0, 4.0]) == (OK, [float(np.float32(0.6)), float(np.float32(0.8))])
    assert rt.call("normalize_l2", [0.0, 0.0]) == (OK, [0.0, 0.0])
    words = rt.call("compress_sign_bits", FloatList([(-1.0) ** (i % 3) for i in range(130)]))
    assert words == [int(w) for w in oracle_mod.compress_sign_bits([(-1.0) ** (i % 3) for i in range(130)])]


def test_ragged_batches_and_bulk_binary(rt, oracle_mod):
    """insert_many with rows of different lengths is rejected as a whole (flat.rs:182-196);
    flat_load_binary takes the same rows as one native-endian f32 binary."""
    ref = rt.call("flat_new", 2, [0])
    assert rt.call("flat_insert_many", ref, [(b"a", [1.0, 0.0]), (b"b", [1.0])]) == (ERROR, b"dimension mismatch")
    assert ok(rt.call("flat_search", ref, [1.0, 0.0], 5)) == []
    assert rt.call("flat_insert_many", ref, [(b"a", [1.0, 0.0]), (b"b", [float("nan") if False else 1e30, 1.0])]) == UNIT
    rng = np.random.default_rng(3)
    n, d = 300, 24
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    ids = [b"doc-%d" % i for i in range(n)]
    ref2 = rt.call("flat_new", 3, [0])
    assert rt.call("flat_load_binary", ref2, ids, x.tobytes(), d) == UNIT
    with pytest.raises(ArgumentError):
        rt.call("flat_load_binary", ref2, ids, x.tobytes()[:-4], d)                # size mismatch
    with pytest.raises(ArgumentError):
        rt.call("flat_load_binary", ref2, ids[:-1] + [7], x.tobytes(), d)
    bad = x.copy()
    bad[17, 3] = np.inf
    assert rt.call("flat_load_binary", ref2, ids, bad.tobytes(), d) == (ERROR, b"vector contains a non-finite value")
    assert rt.call("flat_load_binary", ref2, ids[:2], x[:2, :5].copy().tobytes(), 5) == (ERROR, b"dimension mismatch")
    packed = oracle_mod.pack_ids(ids)
    q = rng.uniform(-1, 1, d).astype(np.float32)
    assert bits(ok(rt.call("flat_search", ref2, FloatList(q), 7))) == bits(oracle_mod.matrix_search(3, x, packed, q, 7))
    ref.release()
    ref2.release()


def test_oracle_parity_5000x384_and_the_staged_searches(rt, oracle_mod):
    """One oracle comparison at config-1 size through the shim, every search-shaped NIF on the same
    index, and each against the Python mirror of Vettore.Nifs (the other user of the C ABI)."""
    from vettore_amd import nifs
    rng = np.random.default_rng(20260721)
    n, d = 5000, 384
    x = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    x[100:140] = x[100]                                                          # a block of identical rows: id order decides
    x = np.stack([oracle_mod.normalize_l2(r) for r in x])
    ids = [b"doc-%d" % (i + 1) for i in range(n)]
    ref = rt.call("flat_new", 2, [0])
    assert rt.call("flat_load_binary", ref, ids, x.tobytes(), d) == UNIT
    mirror = nifs.flat_new_cosine()
    assert nifs.flat_load_matrix(mirror, ids, x) == ("ok", ())
    packed = oracle_mod.pack_ids(ids)
    qs = np.stack([oracle_mod.normalize_l2(q) for q in rng.uniform(-1, 1, (6, d)).astype(np.float32)])
    qs[0] = x[100]
    for q in qs:
        for k in (1, 10, 64):
            got = ok(rt.call("flat_search", ref, FloatList(q), k))
            assert bits(got) == bits(oracle_mod.matrix_search(2, x, packed, q, k))
    batch = ok(rt.call("flat_search_batch", ref, [FloatList(q) for q in qs], 10))
    assert [bits(h) for h in batch] == [bits(oracle_mod.matrix_search(2, x, packed, q, 10)) for q in qs]
    assert rt.call("flat_search_batch", ref, [], 10) == (OK, [])
    assert rt.call("flat_search_batch", ref, [FloatList(qs[0]), [1.0]], 10) == (ERROR, b"dimension mismatch")
    q = qs[1]
    assert bits(ok(rt.call("flat_quantized_search", ref, FloatList(q), 100, 10))) == \
        bits(nifs.flat_quantized_search(mirror, q, 100, 10)[1])
    assert bits(ok(rt.call("flat_funnel_search", ref, FloatList(q), [128], 100, 10))) == \
        bits(nifs.flat_funnel_search(mirror, q, [128], 100, 10)[1])
    n3 = 20_000                                                                  # (enough rows for the grouped Hamming pass)
    x3 = np.stack([oracle_mod.normalize_l2(r) for r in rng.uniform(-1, 1, (n3, 64)).astype(np.float32)])
    ref3 = rt.call("flat_new", 2, [0])
    assert rt.call("flat_load_binary", ref3, [b"g%d" % i for i in range(n3)], x3.tobytes(), 64) == UNIT
    q3 = [FloatList(v) for v in x3[:5]]
    grouped = ok(rt.call("flat_quantized_search_batch", ref3, q3, 100, 10))
    assert [bits(h) for h in grouped] == [bits(ok(rt.call("flat_quantized_search", ref3, v, 100, 10))) for v in q3]
    assert rt.call("flat_quantized_search_batch", ref3, [], 100, 10) == (OK, [])
    assert rt.call("flat_quantized_search_batch", ref3, [q3[0], [1.0]], 100, 10) == (ERROR, b"dimension mismatch")
    grouped = ok(rt.call("flat_funnel_search_batch", ref3, q3, [16, 32], 100, 10))
    assert [bits(h) for h in grouped] == [bits(ok(rt.call("flat_funnel_search", ref3, v, [16, 32], 100, 10))) for v in q3]
    assert rt.call("flat_funnel_search_batch", ref3, [], [16], 100, 10) == (OK, [])
    assert rt.call("flat_funnel_search_batch", ref3, q3, [65], 100, 10) == (ERROR, b"invalid prefix dimensions")
    ref3.release()
    gens = [(0, 50, [64]), (1, 60, []), (2, 20, [])]
    assert bits(ok(rt.call("flat_hybrid_search", ref, FloatList(q), gens, 10))) == \
        bits(nifs.flat_hybrid_search(mirror, q, [(g[0], g[1], g[2]) for g in gens], 10)[1])
    with pytest.raises(ArgumentError):
        rt.call("flat_hybrid_search", ref, FloatList(q), [(0, 50)], 10)
    with pytest.raises(ArgumentError):
        rt.call("flat_funnel_search", ref, FloatList(q), [128.0], 100, 10)
    ref.release()


def test_resource_destructor_frees_the_index(rt):
    """The reference's ResourceArc drops the index with the last reference; here the destructor
    must run exactly once, when the last term holding the resource goes, and give the HBM back."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")                                               # (the copy the library already mapped)

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    n, d = 400_000, 256                                                          # 410 MB of rows
    before_live, before_dtor = rt.live_resources(), rt.dtor_calls()
    free0 = free_bytes()
    ref = rt.call("flat_new", 0, [0])
    second = nif_runtime.Resource(rt, ref.term)                                  # another process holding the same reference
    rows = np.zeros((n, d), np.float32)
    rows[:, 0] = np.arange(n)
    assert rt.call("flat_load_binary", ref, [b"%d" % i for i in range(n)], rows.tobytes(), d) == UNIT
    assert free_bytes() < free0 - 300 * 2 ** 20
    assert rt.live_resources() == before_live + 1
    ref.release()
    assert rt.dtor_calls() == before_dtor                                        # still referenced
    assert ok(rt.call("flat_search", second, FloatList([3.0] + [0.0] * (d - 1)), 1)) == [(b"3", 0.0)]
    second.release()
    assert rt.dtor_calls() == before_dtor + 1 and rt.live_resources() == before_live
    assert free_bytes() > free0 - 64 * 2 ** 20


def test_multi_device_reference_is_one_resource(rt, oracle_mod):
    """flat_new(code, [0, 0, 0]): three shards behind ONE reference (all on device 0 here)."""
    ref = rt.call("flat_new", 0, [0, 0, 0])
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (2000, 16)).astype(np.float32)
    ids = [b"k%d" % i for i in range(2000)]
    assert rt.call("flat_load_binary", ref, ids, x.tobytes(), 16) == UNIT
    q = rng.uniform(-1, 1, 16).astype(np.float32)
    assert bits(ok(rt.call("flat_search", ref, FloatList(q), 9))) == bits(oracle_mod.matrix_search(0, x, oracle_mod.pack_ids(ids), q, 9))
    ref.release()
