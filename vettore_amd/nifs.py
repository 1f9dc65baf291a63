"""Python mirror of `Vettore.Nifs` for the flat-index hot path
(/root/reference/lib/vettore_nifs.ex:58-174, native/vettore/src/nifs.rs).

Same function names, argument meaning and return shapes as the Elixir stubs,
with Elixir terms spelled as Python values:

    {:ok, value}        -> ("ok", value)
    {:ok, {}}           -> ("ok", ())          (Rustler's encoding of Ok(()))
    {:error, "string"}  -> ("error", "string") (the reference's exact strings)
    bare reference      -> FlatRef

Every call goes through the C ABI of libvettore_hip.so into HIP kernels; badly
typed arguments raise (the NIF's ArgumentError / badarg).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Sequence, Tuple

import numpy as np

from . import _lib

DEVICE = 0  # HIP device ordinal used by flat_new_* and the stateless helpers
USIZE_MAX = (1 << 64) - 1
_F32_MAX = float(np.finfo(np.float32).max)

METRICS = [
    "l2", "l2_squared", "cosine", "inner_product", "negative_inner_product",
    "manhattan", "chebyshev", "hamming", "jaccard",
]
METRIC_CODE = {name: i for i, name in enumerate(METRICS)}


def set_device(device: int):
    global DEVICE
    DEVICE = int(device)


class FlatRef:
    """The `reference()` returned by flat_new_*: owns a vt_flat handle; the
    finalizer plays the role of the ResourceArc destructor."""

    def __init__(self, handle, metric: int):
        self._h = handle
        self.metric = metric

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:   # (module globals are None while the interpreter shuts down)
            _lib.load().vt_flat_free(h)

    @property
    def handle(self):
        if not self._h:
            raise ValueError("flat index already freed")
        return self._h

    def __len__(self):
        return _lib.load().vt_flat_len(self.handle)

    @property
    def dimension(self):
        d = _lib.load().vt_flat_dimension(self.handle)
        return None if d < 0 else d


def _bytes(x) -> bytes:
    if isinstance(x, str):
        return x.encode()
    if isinstance(x, (bytes, bytearray)):
        return bytes(x)
    raise TypeError("badarg: id must be a binary")


def _f32_list(v) -> np.ndarray:
    """Rustler decodes [float] into Vec<f32>: non-floats and doubles outside
    the f32 range are badarg; NaN/inf doubles narrow to f32 NaN/inf."""
    a = np.asarray(v, dtype=np.float64).reshape(-1) if not isinstance(v, np.ndarray) else v.reshape(-1)
    if a.dtype != np.float32:
        a64 = np.asarray(a, dtype=np.float64)
        finite = np.isfinite(a64)
        if np.any(np.abs(a64[finite]) > _F32_MAX):
            raise TypeError("badarg: float out of f32 range")
        with np.errstate(over="ignore"):
            a = a64.astype(np.float32)
    return np.ascontiguousarray(a)


def _u64_list(v) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.uint64).reshape(-1))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _szp(a):
    return a.ctypes.data_as(C.POINTER(C.c_size_t))


def _pack_ids(ids: Iterable) -> Tuple[bytes, np.ndarray]:
    bs = [_bytes(i) for i in ids]
    off = np.zeros(len(bs) + 1, dtype=np.uintp)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    return b"".join(bs), off


def _pack_ragged(rows: Sequence[np.ndarray], dtype) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(rows) + 1, dtype=np.uintp)
    if rows:
        off[1:] = np.cumsum([r.size for r in rows])
    vals = np.ascontiguousarray(np.concatenate(rows)) if rows and off[-1] else np.zeros(0, dtype)
    return vals.astype(dtype, copy=False), off


def _err(status: int):
    return ("error", _lib.error_text(status))


def _export_hits(h, with_keys: bool):
    """One bulk copy across the ABI, then Python slicing."""
    L = _lib.load()
    try:
        n = L.vt_hits_len(h)
        if n == 0:
            return []
        blob = C.create_string_buffer(max(1, L.vt_hits_id_bytes(h)))
        off = (C.c_size_t * (n + 1))()
        raw = (C.c_float * n)()
        keys = (C.c_uint32 * n)() if with_keys else None
        L.vt_hits_export(h, blob, off, raw, keys)
        data = blob.raw
        if with_keys:
            return [(data[off[i]:off[i + 1]], float(raw[i]), int(keys[i])) for i in range(n)]
        return [(data[off[i]:off[i + 1]], float(raw[i])) for i in range(n)]
    finally:
        L.vt_hits_free(h)


def _take_hits(h) -> List[Tuple[bytes, float]]:
    return _export_hits(h, False)


def _take_hits_with_keys(h):
    return _export_hits(h, True)


# ----------------------------------------------------------------- flat_new_*
def _flat_new(metric: int) -> FlatRef:
    h = C.c_void_p()
    st = _lib.load().vt_flat_new(metric, DEVICE, C.byref(h))
    if st != 0:
        raise RuntimeError("flat_new: " + _lib.error_text(st))
    return FlatRef(h, metric)


def flat_new_sharded(metric: int, devices: Sequence[int]) -> FlatRef:
    """One resource over several GPUs of the node (vt_flat_new_sharded): same handle, same
    calls, rows dealt to the shards by a hash of their id."""
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    h = C.c_void_p()
    st = _lib.load().vt_flat_new_sharded(metric, devs, len(devices), C.byref(h))
    if st != 0:
        raise RuntimeError("flat_new_sharded: " + _lib.error_text(st))
    return FlatRef(h, metric)


def flat_shard_count(index: FlatRef) -> int:
    return int(_lib.load().vt_flat_shard_count(index.handle))


def flat_shard_lens(index: FlatRef) -> List[int]:
    L = _lib.load()
    return [int(L.vt_flat_shard_len(index.handle, s)) for s in range(flat_shard_count(index))]


def flat_coalesce_stats(index: FlatRef):
    """(batches, searches they carried) of the searches that met on this handle (vt_flat_coalesce_stats)."""
    a, b = C.c_uint64(), C.c_uint64()
    st = _lib.load().vt_flat_coalesce_stats(index.handle, C.byref(a), C.byref(b))
    if st != 0:
        raise RuntimeError("flat_coalesce_stats: " + _lib.error_text(st))
    return int(a.value), int(b.value)


def flat_shard_memory(index: FlatRef, shard: int = 0):
    """(row capacity, slab bytes, mapped chunks) of a shard (vt_flat_shard_memory)."""
    cap, nbytes, chunks = C.c_size_t(), C.c_size_t(), C.c_size_t()
    st = _lib.load().vt_flat_shard_memory(index.handle, shard, C.byref(cap), C.byref(nbytes), C.byref(chunks))
    if st != 0:
        raise RuntimeError("flat_shard_memory: " + _lib.error_text(st))
    return int(cap.value), int(nbytes.value), int(chunks.value)


def flat_route_ids(index: FlatRef, ids_packed: Tuple[bytes, np.ndarray]) -> np.ndarray:
    """Shard of every id of a packed id batch."""
    blob, off = ids_packed
    off = np.ascontiguousarray(off, dtype=np.uintp)
    out = np.zeros(len(off) - 1, dtype=np.uint32)
    st = _lib.load().vt_flat_route_ids(index.handle, len(off) - 1, blob, _szp(off), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    if st != 0:
        raise RuntimeError(_lib.error_text(st))
    return out


def flat_set_exchange(index: FlatRef, mode: int):
    st = _lib.load().vt_flat_set_exchange(index.handle, mode)
    return "ok" if st == 0 else _err(st)


def flat_exchange(index: FlatRef) -> int:
    return int(_lib.load().vt_flat_exchange(index.handle))


def flat_exchange_note(index: FlatRef) -> str:
    """Which exchange a multi-shard handle chose at creation, and why RCCL was refused if it was."""
    return (_lib.load().vt_flat_exchange_note(index.handle) or b"").decode()


def flat_rccl_ranks(index: FlatRef) -> int:
    return int(_lib.load().vt_flat_rccl_ranks(index.handle))


def flat_new_l2(): return _flat_new(0)                       # nifs.rs:200-204
def flat_new_l2_squared(): return _flat_new(1)               # nifs.rs:206-210
def flat_new_cosine(): return _flat_new(2)                   # nifs.rs:212-216
def flat_new_inner_product(): return _flat_new(3)            # nifs.rs:218-222
def flat_new_negative_inner_product(): return _flat_new(4)   # nifs.rs:224-228
def flat_new_manhattan(): return _flat_new(5)                # nifs.rs:230-234
def flat_new_chebyshev(): return _flat_new(6)                # nifs.rs:236-240
def flat_new_hamming(): return _flat_new(7)                  # nifs.rs:242-246
def flat_new_jaccard(): return _flat_new(8)                  # nifs.rs:248-252


def flat_insert(index: FlatRef, id_, vector):
    """nifs.rs:259-271."""
    b, v = _bytes(id_), _f32_list(vector)
    st = _lib.load().vt_flat_insert(index.handle, b, len(b), _fp(v), v.size)
    return ("ok", ()) if st == 0 else _err(st)


def flat_insert_many(index: FlatRef, vectors: Sequence[Tuple[object, Sequence[float]]]):
    """nifs.rs:273-284."""
    ids, ioff = _pack_ids(i for i, _ in vectors)
    vals, voff = _pack_ragged([_f32_list(v) for _, v in vectors], np.float32)
    st = _lib.load().vt_flat_insert_many(index.handle, len(vectors), ids, _szp(ioff), _fp(vals), _szp(voff))
    return ("ok", ()) if st == 0 else _err(st)


def flat_delete(index: FlatRef, id_):
    """nifs.rs:286-295."""
    b = _bytes(id_)
    st = _lib.load().vt_flat_delete(index.handle, b, len(b))
    return ("ok", ()) if st == 0 else _err(st)


def flat_search(index: FlatRef, query, limit: int):
    """nifs.rs:297-309 -> [(id, raw)] ascending by (rank, id)."""
    if not isinstance(limit, int) or limit < 0 or limit > USIZE_MAX:
        raise TypeError("badarg: limit must fit usize")
    q = _f32_list(query)
    h = C.c_void_p()
    st = _lib.load().vt_flat_search(index.handle, _fp(q), q.size, limit, C.byref(h))
    return ("ok", _take_hits(h)) if st == 0 else _err(st)


def flat_search_batch(index: FlatRef, queries, limit: int):
    """Extension: `queries` is an [nq][d] matrix; returns ("ok", [hits per query]),
    each list identical to flat_search of that query."""
    q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32))
    if q.ndim != 2:
        raise TypeError("badarg: queries must be a matrix")
    nq, d = q.shape
    outs = (C.c_void_p * max(nq, 1))()
    st = _lib.load().vt_flat_search_batch(index.handle, _fp(q.reshape(-1)), nq, d, limit, outs)
    if st != 0:
        return _err(st)
    return ("ok", [_take_hits(C.c_void_p(outs[i])) for i in range(nq)])


def flat_search_packed(index: FlatRef, query, limit: int, records: np.ndarray):
    """flat_search whose hits land as 64-byte wire records (vt_hits_pack) in the
    caller's uint8 array [>= limit][64]; returns ("ok", (count, any_long_id)) with
    one C call for the search and one for the serialisation."""
    q = _f32_list(query)
    L = _lib.load()
    h = C.c_void_p()
    st = L.vt_flat_search(index.handle, _fp(q), q.size, limit, C.byref(h))
    if st != 0:
        return _err(st)
    try:
        cap = records.shape[0]
        n = L.vt_hits_pack(h, records.ctypes.data_as(C.c_void_p), cap)
        lens = records[:n, 8:12].view(np.uint32).reshape(-1)
        long_ids = None
        if n and int(lens.max()) > 52:  # ids that do not fit a record travel separately
            ln = C.c_size_t()
            long_ids = [C.string_at(L.vt_hits_id(h, i, C.byref(ln)), ln.value) for i in range(n)]
    finally:
        L.vt_hits_free(h)
    return ("ok", (int(n), long_ids))


def flat_search_batch_blocks(index: FlatRef, queries, limit: int, blocks: np.ndarray):
    """flat_search_batch whose hit lists land as wire blocks (vt_hits_pack_many) in the caller's
    uint8 array [nq][limit + 1][64]: record 0 of a block is its header {u32 count, u32 long_ids}.
    Returns ("ok", long_ids) where long_ids is None or, when some id does not fit a record, the
    ids of every hit as [[bytes] per query]."""
    q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32))
    if q.ndim != 2:
        raise TypeError("badarg: queries must be a matrix")
    nq, d = q.shape
    assert blocks.dtype == np.uint8 and blocks.shape == (nq, limit + 1, 64) and blocks.flags.c_contiguous
    L = _lib.load()
    outs = (C.c_void_p * max(nq, 1))()
    st = L.vt_flat_search_batch(index.handle, _fp(q.reshape(-1)), nq, d, limit, outs)
    if st != 0:
        return _err(st)
    try:
        L.vt_hits_pack_many(outs, nq, limit, blocks.ctypes.data_as(C.c_void_p))
        long_ids = None
        if nq and int(blocks[:, 0, 4:8].view(np.uint32).max()) != 0:
            ln = C.c_size_t()
            long_ids = [[C.string_at(L.vt_hits_id(C.c_void_p(outs[i]), j, C.byref(ln)), ln.value)
                         for j in range(L.vt_hits_len(C.c_void_p(outs[i])))] for i in range(nq)]
    finally:
        for i in range(nq):
            L.vt_hits_free(C.c_void_p(outs[i]))
    return ("ok", long_ids)


def hit_blocks_merge(blocks: np.ndarray, world: int, nq: int, limit: int) -> np.ndarray:
    """[world][nq][limit + 1][64] gathered wire blocks -> [nq][limit + 1][64]: per query the `limit` best
    over all ranks by (rank key, id bytes) (vt_hit_blocks_merge; FlatHit::cmp, flat.rs:34-40)."""
    b = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(world, nq, limit + 1, 64)
    out = np.zeros((nq, limit + 1, 64), dtype=np.uint8)
    st = _lib.load().vt_hit_blocks_merge(b.ctypes.data_as(C.c_void_p), world, nq, limit, out.ctypes.data_as(C.c_void_p))
    if st != 0:
        raise RuntimeError(_lib.error_text(st))
    return out


def unpack_block(block: np.ndarray):
    """[(id, raw)] of one wire block ([limit + 1][64]); ids longer than 52 bytes come back as None."""
    head = block[0, :8].view(np.uint32)
    return [(h[0], h[1]) for h in unpack_records(block[1:], int(head[0]))]


def unpack_records(records: np.ndarray, count: int):
    """[(id, raw, rank_key)] from `count` wire records (ids longer than 52 bytes come back as None)."""
    out = []
    for i in range(count):
        rec = records[i]
        w = rec[:12].view(np.uint32)
        n = int(w[2])
        out.append((bytes(rec[12:12 + n]) if n <= 52 else None, float(rec[4:8].view(np.float32)[0]), int(w[0])))
    return out


def flat_search_with_keys(index: FlatRef, query, limit: int):
    """flat_search plus each hit's rank sort key (for cross-shard merges)."""
    q = _f32_list(query)
    h = C.c_void_p()
    st = _lib.load().vt_flat_search(index.handle, _fp(q), q.size, limit, C.byref(h))
    return ("ok", _take_hits_with_keys(h)) if st == 0 else _err(st)


# -------------------------------------------------------- stateless helpers
def vector_top_k(vectors, query, metric_code: int, dimensions: int, limit: int):
    """nifs.rs:151-162."""
    if not isinstance(metric_code, int) or not 0 <= metric_code <= 255:
        raise TypeError("badarg: metric_code is a u8")
    ids, ioff = _pack_ids(i for i, _ in vectors)
    vals, voff = _pack_ragged([_f32_list(v) for _, v in vectors], np.float32)
    q = _f32_list(query)
    h = C.c_void_p()
    st = _lib.load().vt_vector_top_k(DEVICE, len(vectors), ids, _szp(ioff), _fp(vals), _szp(voff), _fp(q), q.size,
                                     metric_code, dimensions, limit, C.byref(h))
    return ("ok", _take_hits(h)) if st == 0 else _err(st)


def binary_top_k(vectors, query, dimensions: int, limit: int):
    """nifs.rs:164-175."""
    ids, ioff = _pack_ids(i for i, _ in vectors)
    vals, voff = _pack_ragged([_u64_list(v) for _, v in vectors], np.uint64)
    q = _u64_list(query)
    h = C.c_void_p()
    st = _lib.load().vt_binary_top_k(DEVICE, len(vectors), ids, _szp(ioff), _up(vals), _szp(voff), _up(q), q.size,
                                     dimensions, limit, C.byref(h))
    return ("ok", _take_hits(h)) if st == 0 else _err(st)


def normalize_l2(vector):
    """nifs.rs:107-111."""
    v = _f32_list(vector)
    out = np.empty_like(v)
    st = _lib.load().vt_normalize_l2(DEVICE, 1, v.size, _fp(v), _fp(out))
    return ("ok", out) if st == 0 else _err(st)


def compress_sign_bits(vector):
    """nifs.rs:125-129: bare list of u64 words."""
    v = _f32_list(vector)
    words = np.zeros((v.size + 63) // 64, dtype=np.uint64)
    st = _lib.load().vt_compress_sign_bits(DEVICE, 1, v.size, _fp(v), _up(words))
    if st != 0:
        raise RuntimeError("compress_sign_bits: " + _lib.error_text(st))
    return [int(w) for w in words]


# -------------------------------------------- bulk / device-side extensions
def flat_load_matrix(index: FlatRef, ids: Sequence, matrix: np.ndarray):
    """insert_many of equal-length rows from one dense host matrix."""
    m = np.ascontiguousarray(matrix, dtype=np.float32)
    idb, ioff = _pack_ids(ids)
    st = _lib.load().vt_flat_load_matrix(index.handle, m.shape[0], m.shape[1], idb, _szp(ioff), _fp(m.reshape(-1)))
    return ("ok", ()) if st == 0 else _err(st)


def flat_load_device_matrix(index: FlatRef, ids_packed: Tuple[bytes, np.ndarray], device_ptr: int, count: int, d: int):
    """insert_many of `count` rows already resident in this device's HBM
    (row-major f32 [count][d] at `device_ptr`)."""
    idb, ioff = ids_packed
    st = _lib.load().vt_flat_load_device_matrix(index.handle, count, d, idb, _szp(ioff), C.c_void_p(device_ptr))
    return ("ok", ()) if st == 0 else _err(st)


def flat_quantized_search(index: FlatRef, query, candidates: int, limit: int):
    """collection.ex:276-295 as one native call on the resident corpus."""
    q = _f32_list(query)
    h = C.c_void_p()
    st = _lib.load().vt_flat_quantized_search(index.handle, _fp(q), q.size, candidates, limit, C.byref(h))
    return ("ok", _take_hits(h)) if st == 0 else _err(st)


def flat_quantized_search_batch(index: FlatRef, queries, candidates: int, limit: int):
    """Extension: nq quantized searches in one call ([nq][d] matrix); each hit list identical to
    flat_quantized_search of that query -- groups of up to eight share a sweep of the sign bits."""
    q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32))
    if q.ndim != 2:
        raise TypeError("badarg: queries must be a matrix")
    nq, d = q.shape
    outs = (C.c_void_p * max(nq, 1))()
    st = _lib.load().vt_flat_quantized_search_batch(index.handle, _fp(q.reshape(-1)), nq, d, candidates, limit, outs)
    if st != 0:
        return _err(st)
    return ("ok", [_take_hits(C.c_void_p(outs[i])) for i in range(nq)])


def flat_funnel_search(index: FlatRef, query, stages: Sequence[int], candidates: int, limit: int):
    """collection.ex:245-260 as one native call on the resident corpus."""
    q = _f32_list(query)
    st = np.ascontiguousarray(np.asarray(list(stages), dtype=np.uintp))
    h = C.c_void_p()
    rc = _lib.load().vt_flat_funnel_search(index.handle, _fp(q), q.size, _szp(st), st.size, candidates, limit, C.byref(h))
    return ("ok", _take_hits(h)) if rc == 0 else _err(rc)


def flat_funnel_search_batch(index: FlatRef, queries, stages: Sequence[int], candidates: int, limit: int):
    """Extension: nq funnel searches with one set of stages in one call ([nq][d] matrix); each hit list
    identical to flat_funnel_search of that query -- on a cosine collection groups of up to eight
    share the stage-1 sweep of the prefixes."""
    q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32))
    if q.ndim != 2:
        raise TypeError("badarg: queries must be a matrix")
    nq, d = q.shape
    st = np.ascontiguousarray(np.asarray(list(stages), dtype=np.uintp))
    outs = (C.c_void_p * max(nq, 1))()
    rc = _lib.load().vt_flat_funnel_search_batch(index.handle, _fp(q.reshape(-1)), nq, d, _szp(st), st.size, candidates, limit, outs)
    if rc != 0:
        return _err(rc)
    return ("ok", [_take_hits(C.c_void_p(outs[i])) for i in range(nq)])


GEN_FUNNEL, GEN_QUANTIZED, GEN_SEARCH = 0, 1, 2


def flat_hybrid_search(index: FlatRef, query, generators, limit: int):
    """hybrid_search(rerank: :exact) on the resident corpus.  `generators` is a list of
    (kind, candidates, stages) with kind in GEN_*; stages only for GEN_FUNNEL."""
    q = _f32_list(query)
    kinds = (C.c_int * len(generators))(*[g[0] for g in generators])
    cands = np.ascontiguousarray([g[1] for g in generators], dtype=np.uintp)
    flat_stages, off = [], [0]
    for g in generators:
        flat_stages.extend(g[2] if g[0] == GEN_FUNNEL else [])
        off.append(len(flat_stages))
    st = np.ascontiguousarray(flat_stages if flat_stages else [0], dtype=np.uintp)
    so = np.ascontiguousarray(off, dtype=np.uintp)
    h = C.c_void_p()
    rc = _lib.load().vt_flat_hybrid_search(index.handle, _fp(q), q.size, kinds, _szp(cands), _szp(so), _szp(st),
                                           len(generators), limit, C.byref(h))
    return ("ok", _take_hits(h)) if rc == 0 else _err(rc)


def rank_ids(ids_packed: Tuple[bytes, np.ndarray]) -> np.ndarray:
    """Position of every id in the bytewise order of all of them (vt_rank_ids)."""
    blob, off = ids_packed
    n = len(off) - 1
    out = np.empty(n, dtype=np.uint32)
    st = _lib.load().vt_rank_ids(blob, _szp(off), n, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    if st != 0:
        raise RuntimeError("rank_ids: " + _lib.error_text(st))
    return out


def flat_set_id_ranks(index: FlatRef, ranks: np.ndarray):
    r = np.ascontiguousarray(ranks, dtype=np.uint32)
    st = _lib.load().vt_flat_set_id_ranks(index.handle, r.ctypes.data_as(C.POINTER(C.c_uint32)), r.size)
    return "ok" if st == 0 else _err(st)


def flat_stream(index: FlatRef) -> int:
    """The hipStream_t (as an integer) the index enqueues its kernels on."""
    return int(_lib.load().vt_flat_stream(index.handle) or 0)


def flat_search_begin(index: FlatRef, query, limit: int, device_block_ptr: int):
    """Enqueue one shard's search; the result block lands at `device_block_ptr`."""
    q = _f32_list(query)
    st = _lib.load().vt_flat_search_begin(index.handle, _fp(q), q.size, limit, C.c_void_p(device_block_ptr))
    return "ok" if st == 0 else _err(st)


class MergeBuffers:
    """Reusable output arrays of flat_merge_gathered."""

    def __init__(self, cap: int = 256):
        self.keys = np.zeros(cap, dtype=np.uint64)
        self.rows = np.zeros(cap, dtype=np.uint32)
        self.raw = np.zeros(cap, dtype=np.float32)
        self.shard = np.zeros(cap, dtype=np.uint32)
        self.count = C.c_size_t()
        self.ptrs = (self.keys.ctypes.data_as(C.POINTER(C.c_uint64)), self.rows.ctypes.data_as(C.POINTER(C.c_uint32)),
                     self.raw.ctypes.data_as(C.POINTER(C.c_float)), self.shard.ctypes.data_as(C.POINTER(C.c_uint32)))


def flat_merge_gathered(index: FlatRef, device_blocks_ptr: int, world: int, limit: int, block_bytes: int,
                        bufs: MergeBuffers):
    """Merge `world` gathered shard blocks on the device, wait, return the number of winners
    (their keys / rows / raw / shard are in `bufs`)."""
    st = _lib.load().vt_flat_merge_gathered(index.handle, C.c_void_p(device_blocks_ptr), world, limit, block_bytes,
                                            *bufs.ptrs, C.byref(bufs.count))
    return ("ok", int(bufs.count.value)) if st == 0 else _err(st)


def flat_set_reduce_order(index: FlatRef, order: int):
    st = _lib.load().vt_flat_set_reduce_order(index.handle, order)
    return "ok" if st == 0 else _err(st)


def flat_set_batch_nominate(index: FlatRef, mode: int):
    """Which matrix-core pass nominates batch candidates: _lib.NOMINATE_BF16 (default) or NOMINATE_F32."""
    st = _lib.load().vt_flat_set_batch_nominate(index.handle, mode)
    return "ok" if st == 0 else _err(st)


def flat_batch_nominate(index: FlatRef) -> int:
    return int(_lib.load().vt_flat_batch_nominate(index.handle))


def flat_set_batch_shadow(index: FlatRef, mode: int):
    """Whether the bf16 nomination pass may keep a bf16 shadow of the rows: _lib.SHADOW_AUTO (default) or SHADOW_OFF."""
    st = _lib.load().vt_flat_set_batch_shadow(index.handle, mode)
    return ("ok", ()) if st == _lib.VT_OK else ("error", _lib.error_text(st))


def flat_set_single_nominate(index: FlatRef, enabled: bool):
    """Opt-in: lone flat_search calls go through the bf16 shadow like a batch of one (same hits, ~0.6 of the scan's time)."""
    st = _lib.load().vt_flat_set_single_nominate(index.handle, 1 if enabled else 0)
    return ("ok", ()) if st == _lib.VT_OK else ("error", _lib.error_text(st))


def flat_batch_shadow(index: FlatRef) -> str:
    """State of shard 0's shadow: "off", "none" (not built yet), "current", "stale" or "refused" (no room)."""
    return _lib.SHADOW_STATE.get(int(_lib.load().vt_flat_batch_shadow(index.handle)), "?")


def set_default_reduce_order(order: int):
    st = _lib.load().vt_set_default_reduce_order(order)
    return "ok" if st == 0 else _err(st)


def debug_set(name: str, value: int):
    """A library setting by name (vt_debug_set, include/vettore_flat.h): the VT_* variables are read once,
    when the library is loaded; afterwards -- and for the switches that have no variable ("force_batch_mfma"
    ...) -- this is the way in.  Process-wide; tests and probes only."""
    st = _lib.load().vt_debug_set(name.encode(), int(value))
    if st != 0:
        raise KeyError("libvettore_hip has no setting %r" % name)


def debug_get(name: str) -> int:
    v = C.c_long()
    if _lib.load().vt_debug_get(name.encode(), C.byref(v)) != 0:
        raise KeyError("libvettore_hip has no setting %r" % name)
    return int(v.value)


class debug_setting:
    """`with nifs.debug_setting("batch_no_mfma", 1): ...` -- the old value comes back afterwards."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = debug_get(self.name)
        debug_set(self.name, self.value)
        return self

    def __exit__(self, *exc):
        debug_set(self.name, self.old)
        return False


def device_read_peak(device: int = 0, nbytes: int = 8 << 30, reps: int = 5):
    """GB/s of the plainest read-only streaming kernel on this box (diagnostic, bench.py)."""
    g = C.c_double()
    st = _lib.load().vt_device_read_peak(device, nbytes, reps, C.byref(g))
    return ("ok", float(g.value)) if st == 0 else _err(st)


def flat_set_profiling(index: FlatRef, enabled: bool):
    _lib.load().vt_flat_set_profiling(index.handle, 1 if enabled else 0)


def flat_get_profile(index: FlatRef, reset: bool = False) -> dict:
    p = _lib.Profile()
    _lib.load().vt_flat_get_profile(index.handle, C.byref(p), 1 if reset else 0)
    return {name: getattr(p, name) for name, _ in _lib.Profile._fields_}


pack_ids = _pack_ids
