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


    # ---- the collection's staged searches on the resident corpus (flat_gpu.ex: quantized_search/3, funnel_search/3,
    # hybrid_search/3 and their batch forms).  The collection has checked the option KEYS (collection.ex:237, :267,
    # :330) and dispatches here when the index module has the function; validation order
    # and error atoms are do_quantized_search / do_funnel_search / do_hybrid_search's.
    @staticmethod
    def _limit_and_candidates(opts):
        limit = opts.get("limit", 10)
        if not _valid_size(limit):
            return ("error", "invalid_limit")                                  # collection.ex:882-887
        candidates = opts.get("candidates", max(limit * 10, limit))           # collection.ex:510
        if not (_valid_size(candidates) and candidates >= limit):
            return ("error", "invalid_candidates")                             # collection.ex:889-895
        return ("ok", limit, candidates)

    @staticmethod
    def _stages(collection, opts):
        if "stages" in opts:                                                   # collection.ex:660-672
            stages = opts["stages"]
        elif "dimensions" in opts:
            stages = [opts["dimensions"]]
        else:
            stages = [min(collection.dimensions, 128)]
        if not (isinstance(stages, list) and stages and all(_valid_size(s) and s <= collection.dimensions for s in stages)):
            return ("error", "invalid_stages")                                 # collection.ex:905-913
        return ("ok", stages)

    @staticmethod
    def _prepare_all(collection, queries):
        out = []
        for q in queries:
            p = collection.prepare_query(q)
            if p[0] != "ok":
                return p
            out.append(p[1])
        return ("ok", out)

    @staticmethod
    def search_batch(collection, queries, opts=None):
        opts = {} if opts is None else opts
        if not isinstance(opts, dict) or any(k != "limit" for k in opts):
            return ("error", "invalid_search_options")
        limit = opts.get("limit", 10)
        if not _valid_size(limit):
            return ("error", "invalid_limit")
        prepared = FlatGpu._prepare_all(collection, queries)
        if prepared[0] != "ok":
            return prepared
        res = nifs.flat_search_batch(collection.index_state, prepared[1], limit)
        if res[0] != "ok":
            return res
        return ("ok", [_hydrate(collection, hits) for hits in res[1]])

    @staticmethod
    def quantized_search(collection, query, opts=None):
        """collection.ex:276-295 on the resident corpus."""
        lc = FlatGpu._limit_and_candidates({} if opts is None else opts)
        if lc[0] != "ok":
            return lc
        q = collection.prepare_query(query)
        if q[0] != "ok":
            return q
        res = nifs.flat_quantized_search(collection.index_state, q[1], lc[2], lc[1])
        return res if res[0] != "ok" else ("ok", _hydrate(collection, res[1]))

    @staticmethod
    def quantized_search_batch(collection, queries, opts=None):
        lc = FlatGpu._limit_and_candidates({} if opts is None else opts)
        if lc[0] != "ok":
            return lc
        prepared = FlatGpu._prepare_all(collection, queries)
        if prepared[0] != "ok":
            return prepared
        res = nifs.flat_quantized_search_batch(collection.index_state, prepared[1], lc[2], lc[1])
        return res if res[0] != "ok" else ("ok", [_hydrate(collection, hits) for hits in res[1]])

    @staticmethod
    def funnel_search(collection, query, opts=None):
        """collection.ex:245-260 on the resident corpus."""
        opts = {} if opts is None else opts
        lc = FlatGpu._limit_and_candidates(opts)
        if lc[0] != "ok":
            return lc
        st = FlatGpu._stages(collection, opts)
        if st[0] != "ok":
            return st
        q = collection.prepare_query(query)
        if q[0] != "ok":
            return q
        res = nifs.flat_funnel_search(collection.index_state, q[1], st[1], lc[2], lc[1])
        return res if res[0] != "ok" else ("ok", _hydrate(collection, res[1]))

    @staticmethod
    def funnel_search_batch(collection, queries, opts=None):
        opts = {} if opts is None else opts
        lc = FlatGpu._limit_and_candidates(opts)
        if lc[0] != "ok":
            return lc
        st = FlatGpu._stages(collection, opts)
        if st[0] != "ok":
            return st
        prepared = FlatGpu._prepare_all(collection, queries)
        if prepared[0] != "ok":
            return prepared
        res = nifs.flat_funnel_search_batch(collection.index_state, prepared[1], st[1], lc[2], lc[1])
        return res if res[0] != "ok" else ("ok", [_hydrate(collection, hits) for hits in res[1]])

    @staticmethod
    def hybrid_search(collection, query, opts=None):
        """collection.ex:325-345 with rerank: exact; generators as run_hybrid_generator takes them (collection.ex:536-556):
        "funnel" | "quantized" | "search", or (name, {options})."""
        opts = {} if opts is None else opts
        limit = opts.get("limit", 10)
        if not _valid_size(limit):
            return ("error", "invalid_limit")
        q = collection.prepare_query(query)
        if q[0] != "ok":
            return q
        generators = opts.get("generators", ["funnel", "quantized"])          # collection.ex:512-513
        if not isinstance(generators, list) or not generators:
            return ("error", "invalid_generators")
        spec = []
        for gen in generators:
            name, gopts = (gen, {}) if isinstance(gen, str) else (gen if isinstance(gen, tuple) and len(gen) == 2 else (None, None))
            if not isinstance(name, str) or not isinstance(gopts, dict):
                return ("error", ("invalid_generator", gen))
            if name not in ("funnel", "quantized", "search", "hnsw"):
                return ("error", ("unknown_generator", name))                  # collection.ex:1142
            allowed = ("candidates", "stages", "dimensions") if name == "funnel" else ("candidates",)
            extra = [k for k in gopts if k not in allowed]
            if extra:
                return ("error", ("unsupported_option", extra[0]))
            if name == "hnsw":
                return ("error", "hnsw_index_required")                        # collection.ex:600
            cand = gopts.get("candidates", max(limit * 10, limit))             # collection.ex:544
            if not _valid_size(cand):
                return ("error", "invalid_candidates")                         # collection.ex:897-902
            if name == "funnel":
                st = FlatGpu._stages(collection, gopts)
                if st[0] != "ok":
                    return st
                spec.append((nifs.GEN_FUNNEL, cand, st[1]))
            else:
                spec.append((nifs.GEN_QUANTIZED if name == "quantized" else nifs.GEN_SEARCH, cand, []))
        if opts.get("rerank", "exact") != "exact":
            return ("error", ("invalid_rerank", opts.get("rerank")))           # (multi-vector rerank: not on the flat path)
        res = nifs.flat_hybrid_search(collection.index_state, q[1], spec, limit)
        return res if res[0] != "ok" else ("ok", _hydrate(collection, res[1]))


def _valid_size(v):
    return isinstance(v, int) and not isinstance(v, bool) and 0 < v <= MAX_NIF_USIZE


def _hydrate(collection, hits):
    out: List[Result] = []
    for id_, raw in hits:
        out.extend(_to_result(collection, id_, raw))
    return out


def _to_result(collection, id_, raw):
    """flat.ex:72-91: hits whose id is no longer in the store are dropped."""
    got = collection.get(id_)
    if got[0] != "ok":
        return []
    emb = got[1]
    score, distance = result_values(collection.metric, raw, collection.score)
    return [Result(id=id_, value=emb.value, score=score, distance=distance, metric=collection.metric,
                   metadata=emb.metadata)]
