"""ctypes wrapper around the CPU ORACLE (oracle/vt_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke(); the product package `vettore_amd` never imports it.

The wrapper mirrors the reference's Rust signatures (native/vettore/src/*.rs):
errors come back as the reference's exact strings wrapped in OracleError.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Iterable, List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvt_oracle.so")

ORDER_PAIR, ORDER_AVX, ORDER_SEQ, ORDER_SSE2 = 0, 1, 2, 3
ORDERS = (ORDER_PAIR, ORDER_AVX, ORDER_SEQ, ORDER_SSE2)
DEFAULT_ORDER = ORDER_SSE2
METRICS = [
    "l2", "l2_squared", "cosine", "inner_product", "negative_inner_product",
    "manhattan", "chebyshev", "hamming", "jaccard",
]
METRIC_CODE = {name: i for i, name in enumerate(METRICS)}
USIZE_MAX = (1 << 64) - 1


class OracleError(Exception):
    """Carries the reference's error string (e.g. "dimension mismatch")."""


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vt_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libvt_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # VT_ORACLE_LIB: another build of the same source (tools/cpu_variants.py times -O3 / -march=native builds)
    path = os.environ.get("VT_ORACLE_LIB") or _LIB_PATH
    if path == _LIB_PATH and not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(path)
    f32p, u64p, szp = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)
    vp = C.c_void_p
    L.vto_strerror.restype = C.c_char_p
    L.vto_strerror.argtypes = [C.c_int]
    L.vto_set_reduce_order.argtypes = [C.c_int]
    L.vto_get_reduce_order.restype = C.c_int
    L.vto_compute.argtypes = [C.c_int, f32p, C.c_size_t, f32p, C.c_size_t, f32p]
    L.vto_compute_checked.argtypes = L.vto_compute.argtypes
    L.vto_rank_value.restype = C.c_float
    L.vto_rank_value.argtypes = [C.c_int, C.c_float]
    L.vto_cosine.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, f32p]
    L.vto_validate_finite.argtypes = [f32p, C.c_size_t]
    L.vto_normalize_l2.argtypes = [f32p, C.c_size_t, f32p]
    L.vto_compress_sign_bits.restype = None
    L.vto_compress_sign_bits.argtypes = [f32p, C.c_size_t, u64p]
    L.vto_packed_hamming.argtypes = [u64p, C.c_size_t, u64p, C.c_size_t, C.c_size_t, f32p]
    L.vto_packed_jaccard.argtypes = L.vto_packed_hamming.argtypes
    L.vto_hits_len.restype = C.c_size_t
    L.vto_hits_len.argtypes = [vp]
    L.vto_hits_id.restype = C.POINTER(C.c_char)
    L.vto_hits_id.argtypes = [vp, C.c_size_t, szp]
    L.vto_hits_raw.restype = C.c_float
    L.vto_hits_raw.argtypes = [vp, C.c_size_t]
    L.vto_hits_free.restype = None
    L.vto_hits_free.argtypes = [vp]
    L.vto_flat_new.restype = vp
    L.vto_flat_new.argtypes = [C.c_int]
    L.vto_flat_free.restype = None
    L.vto_flat_free.argtypes = [vp]
    L.vto_flat_len.restype = C.c_size_t
    L.vto_flat_len.argtypes = [vp]
    L.vto_flat_dimension.restype = C.c_long
    L.vto_flat_dimension.argtypes = [vp]
    L.vto_flat_insert.argtypes = [vp, C.c_char_p, C.c_size_t, f32p, C.c_size_t]
    L.vto_flat_insert_many.argtypes = [vp, C.c_size_t, C.c_char_p, szp, f32p, szp]
    L.vto_flat_delete.restype = None
    L.vto_flat_delete.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.vto_flat_search.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vto_vector_top_k.argtypes = [C.c_size_t, C.c_char_p, szp, f32p, szp, f32p, C.c_size_t,
                                   C.c_int, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vto_binary_top_k.argtypes = [C.c_size_t, C.c_char_p, szp, u64p, szp, u64p, C.c_size_t,
                                   C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vto_matrix_search.argtypes = [C.c_int, f32p, C.c_size_t, C.c_size_t, C.c_char_p, szp,
                                    f32p, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise OracleError(lib().vto_strerror(rc).decode())


def _f32(v) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(-1))


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u64(v) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.uint64).reshape(-1))


def _up(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _bytes(x) -> bytes:
    return x.encode() if isinstance(x, str) else bytes(x)


def pack_ids(ids: Iterable) -> Tuple[bytes, np.ndarray]:
    bs = [_bytes(i) for i in ids]
    off = np.zeros(len(bs) + 1, dtype=np.uintp)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    return b"".join(bs), off


def pack_ragged(rows: Sequence, dtype) -> Tuple[np.ndarray, np.ndarray]:
    arrs = [np.asarray(r, dtype=dtype).reshape(-1) for r in rows]
    off = np.zeros(len(arrs) + 1, dtype=np.uintp)
    if arrs:
        off[1:] = np.cumsum([a.size for a in arrs])
        vals = np.ascontiguousarray(np.concatenate(arrs)) if off[-1] else np.zeros(0, dtype)
    else:
        vals = np.zeros(0, dtype)
    return vals, off


def _szp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_size_t))


def _take_hits(h) -> List[Tuple[bytes, float]]:
    L = lib()
    out = []
    try:
        n = L.vto_hits_len(h)
        ln = C.c_size_t()
        for i in range(n):
            p = L.vto_hits_id(h, i, C.byref(ln))
            out.append((C.string_at(p, ln.value), float(L.vto_hits_raw(h, i))))
    finally:
        L.vto_hits_free(h)
    return out


def set_reduce_order(order: int):
    lib().vto_set_reduce_order(order)


def get_reduce_order() -> int:
    return lib().vto_get_reduce_order()


def compute(metric: int, left, right, checked: bool = False) -> float:
    a, b = _f32(left), _f32(right)
    out = C.c_float()
    fn = lib().vto_compute_checked if checked else lib().vto_compute
    _check(fn(metric, _fp(a), a.size, _fp(b), b.size, C.byref(out)))
    return np.float32(out.value)


def rank_value(metric: int, raw: float) -> np.float32:
    return np.float32(lib().vto_rank_value(metric, C.c_float(raw)))


def cosine(left, right) -> np.float32:
    a, b = _f32(left), _f32(right)
    out = C.c_float()
    _check(lib().vto_validate_finite(_fp(a), a.size))
    _check(lib().vto_validate_finite(_fp(b), b.size))
    _check(lib().vto_cosine(_fp(a), a.size, _fp(b), b.size, C.byref(out)))
    return np.float32(out.value)


def normalize_l2(v) -> np.ndarray:
    a = _f32(v)
    out = np.empty_like(a)
    _check(lib().vto_normalize_l2(_fp(a), a.size, _fp(out)))
    return out


def compress_sign_bits(v) -> np.ndarray:
    a = _f32(v)
    words = np.zeros((a.size + 63) // 64, dtype=np.uint64)
    lib().vto_compress_sign_bits(_fp(a), a.size, _up(words))
    return words


def packed_hamming(left, right, dimensions: int) -> np.float32:
    a, b = _u64(left), _u64(right)
    out = C.c_float()
    _check(lib().vto_packed_hamming(_up(a), a.size, _up(b), b.size, dimensions, C.byref(out)))
    return np.float32(out.value)


def packed_jaccard(left, right, dimensions: int) -> np.float32:
    a, b = _u64(left), _u64(right)
    out = C.c_float()
    _check(lib().vto_packed_jaccard(_up(a), a.size, _up(b), b.size, dimensions, C.byref(out)))
    return np.float32(out.value)


class FlatIndex:
    """flat.rs FlatIndex."""

    def __init__(self, metric: int):
        self._h = lib().vto_flat_new(metric)
        if not self._h:
            raise OracleError("unknown metric")
        self.metric = metric

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().vto_flat_free(h)

    def __len__(self):
        return lib().vto_flat_len(self._h)

    @property
    def dimension(self):
        d = lib().vto_flat_dimension(self._h)
        return None if d < 0 else d

    def insert(self, id_, vector):
        b, v = _bytes(id_), _f32(vector)
        _check(lib().vto_flat_insert(self._h, b, len(b), _fp(v), v.size))

    def insert_many(self, items: Sequence[Tuple[object, Sequence[float]]]):
        ids, ioff = pack_ids(i for i, _ in items)
        vals, voff = pack_ragged([v for _, v in items], np.float32)
        _check(lib().vto_flat_insert_many(self._h, len(items), ids, _szp(ioff), _fp(vals), _szp(voff)))

    def insert_matrix(self, ids: Sequence, matrix: np.ndarray):
        m = np.ascontiguousarray(matrix, dtype=np.float32)
        idb, ioff = pack_ids(ids)
        voff = (np.arange(m.shape[0] + 1, dtype=np.uintp) * m.shape[1]).astype(np.uintp)
        _check(lib().vto_flat_insert_many(self._h, m.shape[0], idb, _szp(ioff), _fp(m.reshape(-1)), _szp(voff)))

    def delete(self, id_):
        b = _bytes(id_)
        lib().vto_flat_delete(self._h, b, len(b))

    def search(self, query, limit: int) -> List[Tuple[bytes, float]]:
        q = _f32(query)
        h = C.c_void_p()
        _check(lib().vto_flat_search(self._h, _fp(q), q.size, limit, C.byref(h)))
        return _take_hits(h)


def vector_top_k(vectors, query, metric_code: int, dimensions: int, limit: int):
    ids, ioff = pack_ids(i for i, _ in vectors)
    vals, voff = pack_ragged([v for _, v in vectors], np.float32)
    q = _f32(query)
    h = C.c_void_p()
    _check(lib().vto_vector_top_k(len(vectors), ids, _szp(ioff), _fp(vals), _szp(voff), _fp(q), q.size,
                                  metric_code, dimensions, limit, C.byref(h)))
    return _take_hits(h)


def binary_top_k(vectors, query, dimensions: int, limit: int):
    ids, ioff = pack_ids(i for i, _ in vectors)
    vals, voff = pack_ragged([v for _, v in vectors], np.uint64)
    q = _u64(query)
    h = C.c_void_p()
    _check(lib().vto_binary_top_k(len(vectors), ids, _szp(ioff), _up(vals), _szp(voff), _up(q), q.size,
                                  dimensions, limit, C.byref(h)))
    return _take_hits(h)


def matrix_search(metric: int, matrix: np.ndarray, ids_packed: Tuple[bytes, np.ndarray], query, limit: int):
    m = np.ascontiguousarray(matrix, dtype=np.float32)
    n, d = m.shape
    idb, ioff = ids_packed
    q = _f32(query)
    h = C.c_void_p()
    _check(lib().vto_matrix_search(metric, _fp(m.reshape(-1)), n, d, idb, _szp(ioff), _fp(q), q.size, limit,
                                   C.byref(h)))
    return _take_hits(h)
