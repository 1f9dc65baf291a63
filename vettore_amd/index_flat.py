"""Python mirror of the reference's index plugin point for the flat path:
the `Vettore.Index` behaviour (/root/reference/lib/vettore/index.ex:12-17) and
its implementation `Vettore.Index.Flat` (lib/vettore/index/flat.ex), backed by
the GPU library instead of the Rust NIF.  This is the module a maintainer would
name `Vettore.Index.FlatGpu` and select with `index: Vettore.Index.FlatGpu`
(INTEGRATION.md).

Return conventions follow Elixir: "ok" | ("ok", value) | ("error", reason).
Atoms are plain strings ("invalid_limit"); native errors are the reference's
error strings ("dimension mismatch").
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List, Optional

from . import nifs

MAX_NIF_USIZE = 4_294_967_295  # flat.ex:13


@dataclass
class Result:
    """Vettore.Result (lib/vettore/result.ex)."""
    id: bytes
    value: Any
    score: float
    distance: Optional[float]
    metric: str
    metadata: Any = None


_NEW = {
    "l2": nifs.flat_new_l2,
    "l2_squared": nifs.flat_new_l2_squared,
    "cosine": nifs.flat_new_cosine,
    "inner_product": nifs.flat_new_inner_product,
    "negative_inner_product": nifs.flat_new_negative_inner_product,
    "manhattan": nifs.flat_new_manhattan,
    "chebyshev": nifs.flat_new_chebyshev,
    "hamming": nifs.flat_new_hamming,
    "jaccard": nifs.flat_new_jaccard,
}

SIMILARITY_METRICS = ("cosine", "inner_product")
DISTANCE_METRICS = ("l2", "l2_squared", "manhattan", "chebyshev", "hamming", "jaccard")


def result_values(metric: str, raw: float, score_mode: str = "raw"):
    """Vettore.Distance.result_values/3 (lib/vettore_distance.ex:98-102, :525-543)."""
    raw = float(raw)
    if metric == "negative_inner_product" and score_mode in ("raw", "similarity"):
        return (-raw, raw)
    if score_mode == "raw" and metric in SIMILARITY_METRICS:
        return (raw, 1.0 - raw if metric == "cosine" else -raw)
    if score_mode == "raw" and metric in DISTANCE_METRICS:
        return (-raw, raw)
    if score_mode == "similarity" and metric in SIMILARITY_METRICS:
        score = (raw + 1.0) / 2.0 if metric == "cosine" else raw
        return (score, 1.0 - raw if metric == "cosine" else -raw)
    if score_mode == "similarity" and metric in DISTANCE_METRICS:
        return (1.0 / (1.0 + raw), raw)
    return (raw, None)


def _normalize_ok(res):
    """flat.ex:93-96."""
    if res == ("ok", ()) or res == "ok":
        return "ok"
    return res


class FlatGpu:
    """`@behaviour Vettore.Index` with callbacks new/2, put/2, put_many/2,
    delete/2, search/3 (flat.ex:15-57)."""

    @staticmethod
    def new(metric: str, opts=None):
        """`index_options` reach new/2 verbatim (collection.ex:100-103).  The built-in flat index
        insists on [] (flat.ex:19-25); this one also takes `device: n` or `devices: [n, ...]`
        -- one resource spread over several GPUs of the node (vt_flat_new_sharded)."""
        opts = {} if opts is None else (dict(opts) if isinstance(opts, (list, dict)) else None)
        if opts is None or any(k not in ("device", "devices") for k in opts) or len(opts) > 1:
            return ("error", "invalid_flat_options")          # flat.ex:19-25
        code = nifs.METRIC_CODE.get(metric)
        if code is None:
            return ("error", ("unsupported_flat_metric", metric))  # flat.ex:69
        devices = opts.get("devices", [opts.get("device", nifs.DEVICE)])
        ok = isinstance(devices, (list, tuple)) and len(devices) > 0 and \
            all(isinstance(d, int) and not isinstance(d, bool) and d >= 0 for d in devices)
        if not ok:
            return ("error", "invalid_flat_options")
        try:
            return ("ok", nifs.flat_new_sharded(code, list(devices)))
        except RuntimeError as e:  # no such device, out of memory: the NIF's {:error, msg}
            return ("error", str(e))

    @staticmethod
    def put(collection, embedding):
        return _normalize_ok(nifs.flat_insert(collection.index_state, embedding.id, embedding.vector))

    @staticmethod
    def put_many(collection, embeddings):
        vectors = [(e.id, e.vector) for e in embeddings]          # flat.ex:35-39
        return _normalize_ok(nifs.flat_insert_many(collection.index_state, vectors))

    @staticmethod
    def delete(collection, id_):
        return _normalize_ok(nifs.flat_delete(collection.index_state, id_))

    @staticmethod
    def search(collection, query, opts=None):
        opts = {} if opts is None else opts
        if not isinstance(opts, dict) or any(k != "limit" for k in opts):
            return ("error", "invalid_search_options")            # flat.ex:105-112
        limit = opts.get("limit", 10)
        if not (isinstance(limit, int) and not isinstance(limit, bool) and 0 < limit <= MAX_NIF_USIZE):
            return ("error", "invalid_limit")                      # flat.ex:98-103
        prepared = collection.prepare_query(query)
        if prepared[0] != "ok":
            return prepared
        res = nifs.flat_search(collection.index_state, prepared[1], limit)
        if res[0] != "ok":
            return res
        out: List[Result] = []
        for id_, raw in res[1]:
            out.extend(_to_result(collection, id_, raw))
        return ("ok", out)


def _to_result(collection, id_, raw):
    """flat.ex:72-91: hits whose id is no longer in the store are dropped."""
    got = collection.get(id_)
    if got[0] != "ok":
        return []
    emb = got[1]
    score, distance = result_values(collection.metric, raw, collection.score)
    return [Result(id=id_, value=emb.value, score=score, distance=distance, metric=collection.metric,
                   metadata=emb.metadata)]
