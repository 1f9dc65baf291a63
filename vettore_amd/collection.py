"""Caller side of the flat hot path: the parts of `Vettore.Collection`
(/root/reference/lib/vettore/collection.ex) that sit directly above the index
plugin -- option defaults, prepare_query, two-phase put with rollback, search
dispatch and quantized_search.  The canonical record store (ETS in the
reference, lib/vettore/store/ets.ex) is a plain dict here: it is out of scope
 and only holds ids, values and metadata.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import numpy as np

from . import nifs
from .index_flat import FlatGpu, Result, result_values, MAX_NIF_USIZE

F32_MAX = 3.4028234663852886e38  # collection.ex:61
METRIC_ALIASES = {"euclidean": "l2", "dot": "inner_product", "dot_product": "inner_product"}  # :1300-1304
METRICS = nifs.METRICS


@dataclass
class Embedding:
    """Vettore.Embedding (lib/vettore_embedding.ex)."""
    id: Any = None
    value: Any = None
    vector: Any = None
    binary_vector: Optional[List[int]] = None
    metadata: Any = None


def _finite_f32(v) -> bool:  # collection.ex:1264-1270
    if isinstance(v, bool) or not isinstance(v, (int, float, np.floating, np.integer)):
        return False
    v = float(v)
    return not math.isnan(v) and -F32_MAX <= v <= F32_MAX


def _validate_vector(vector, dimensions):  # collection.ex:1087-1093
    if not isinstance(vector, (list, tuple, np.ndarray)):
        return ("error", "invalid_vector")
    if len(vector) != dimensions:
        return ("error", "dimension_mismatch")
    if all(_finite_f32(x) for x in vector):
        return "ok"
    return ("error", "invalid_vector")


class Collection:
    """%Vettore.Collection{} for `store: dict`, `index: FlatGpu` (or any object
    with the five Vettore.Index callbacks -- the reference's plugin point,
    collection.ex:72, :1283-1298)."""

    def __init__(self):
        raise TypeError("use Collection.new(...)")

    @classmethod
    def new(cls, dimensions=None, metric="cosine", normalize=None, index="flat", index_options=None,
            score="raw", name=None):
        metric = METRIC_ALIASES.get(metric, metric)
        if not (isinstance(dimensions, int) and dimensions > 0):
            return ("error", "invalid_dimensions")
        if metric not in METRICS:
            return ("error", "invalid_metric")
        if normalize is None:
            normalize = "l2" if metric == "cosine" else "none"     # collection.ex:1317-1319
        if normalize not in ("none", "l2"):
            return ("error", "invalid_normalization")               # zscore/minmax: out of scope
        if score not in ("raw", "similarity"):
            return ("error", "invalid_score_mode")
        index_mod = FlatGpu if index in ("flat", "flat_gpu") else index
        for cb in ("new", "put", "put_many", "delete", "search"):
            if not hasattr(index_mod, cb):
                return ("error", "invalid_index")
        made = index_mod.new(metric, index_options or [])
        if made[0] != "ok":
            return made
        self = object.__new__(cls)
        self.name, self.dimensions, self.metric = name, dimensions, metric
        self.normalize, self.score = normalize, score
        self.index_mod, self.index_state = index_mod, made[1]
        self.store: Dict[bytes, Embedding] = {}
        self.open = True
        return ("ok", self)

    # -- store (ETS stand-in) ------------------------------------------------
    def get(self, id_):
        emb = self.store.get(nifs._bytes(id_))
        return ("ok", emb) if emb is not None else ("error", "not_found")

    def all(self):
        return ("ok", list(self.store.values()))

    def close(self):
        self.open = False
        return "ok"

    # -- collection.ex:352-357 ----------------------------------------------
    def prepare_query(self, query):
        if not self.open:
            return ("error", "closed")
        ok = _validate_vector(query, self.dimensions)
        if ok != "ok":
            return ok
        return self._normalize(query)

    def _normalize(self, vector):
        if self.normalize == "none":
            return ("ok", [float(x) / 1 for x in vector])
        res = nifs.normalize_l2([float(x) for x in vector])    # vettore_distance.ex:62-66
        if res[0] != "ok":
            return ("error", "invalid_vector")
        return ("ok", res[1])

    # -- collection.ex:921-937 ----------------------------------------------
    def _prepare_embedding(self, emb):
        if isinstance(emb, dict):
            emb = Embedding(id=emb.get("id"), value=emb.get("value"), vector=emb.get("vector"),
                            metadata=emb.get("metadata"))
        if not isinstance(emb, Embedding):
            return ("error", "invalid_embedding")
        if not isinstance(emb.id, (str, bytes)) or len(emb.id) == 0:
            return ("error", "missing_id")
        ok = _validate_vector(emb.vector, self.dimensions)
        if ok != "ok":
            return ok
        vec = self._normalize(emb.vector)
        if vec[0] != "ok":
            return vec
        bits = nifs.compress_sign_bits(vec[1])                  # collection.ex:926, :941-946
        idb = nifs._bytes(emb.id)
        return ("ok", Embedding(id=idb, value=emb.value if emb.value is not None else idb, vector=vec[1],
                                binary_vector=bits, metadata=emb.metadata))

    # -- collection.ex:168-189, :459-479 -------------------------------------
    def put(self, emb):
        p = self._prepare_embedding(emb)
        if p[0] != "ok":
            return p
        e = p[1]
        if e.id in self.store:
            return ("error", "duplicate_id")
        self.store[e.id] = e
        res = self.index_mod.put(self, e)
        if res != "ok":
            self._rollback([e])
            return res
        return "ok"

    def put_many(self, embs):
        if not isinstance(embs, list):
            return ("error", "invalid_embeddings")
        prepared = []
        for emb in embs:
            p = self._prepare_embedding(emb)
            if p[0] != "ok":
                return p
            prepared.append(p[1])
        ids = [e.id for e in prepared]
        if len(set(ids)) != len(ids) or any(i in self.store for i in ids):
            return ("error", "duplicate_id")
        for e in prepared:
            self.store[e.id] = e
        res = self.index_mod.put_many(self, prepared)
        if res != "ok":
            self._rollback(prepared)
            return res
        return "ok"

    def _rollback(self, embs):
        for e in embs:
            self.index_mod.delete(self, e.id)
            self.store.pop(e.id, None)

    def delete(self, id_):
        if not isinstance(id_, (str, bytes)):
            return ("error", "invalid_id")
        idb = nifs._bytes(id_)
        res = self.index_mod.delete(self, idb)
        if res == "ok":
            self.store.pop(idb, None)
        return res

    # -- collection.ex:224-228 -----------------------------------------------
    def search(self, query, opts=None):
        opts = {} if opts is None else opts
        if not isinstance(opts, dict):
            return ("error", "invalid_options")
        bad = [k for k in opts if k != "limit"]
        if bad:
            return ("error", ("unsupported_option", bad[0]))
        return self.index_mod.search(self, query, opts)

    # -- collection.ex:234-345: funnel_search / quantized_search / hybrid_search --------------------
    # The reference checks the option keys (validate_options, collection.ex:237, :267, :330) and then runs its own
    # ETS + CPU-NIF composition whatever the index module is.  With the dispatch INTEGRATION.md section 3 adds to
    # collection.ex -- `if function_exported?(collection.index_mod, :quantized_search, 3), do: ...` -- an index module
    # that keeps the corpus resident answers instead.  This mirror has no ETS composition of its own (the store is out
    # of scope): an index module without the function is {:error, :not_supported_by_index} here.
    def _staged(self, name, allowed, query, opts):
        opts = {} if opts is None else opts
        if not isinstance(opts, dict):
            return ("error", "invalid_options")                         # collection.ex:1133
        bad = [k for k in opts if k not in allowed]
        if bad:
            return ("error", ("unsupported_option", bad[0]))            # collection.ex:1125-1126
        fn = getattr(self.index_mod, name, None)
        if fn is None:
            return ("error", "not_supported_by_index")
        return fn(self, query, opts)

    def funnel_search(self, query, opts=None):
        return self._staged("funnel_search", ("limit", "candidates", "stages", "dimensions"), query, opts)   # :56

    def quantized_search(self, query, opts=None):
        return self._staged("quantized_search", ("limit", "candidates"), query, opts)                        # :57

    def hybrid_search(self, query, opts=None):
        return self._staged("hybrid_search", ("limit", "generators", "rerank"), query, opts)                 # :59

    # (extensions of the adapter: lists of queries, one call)
    def search_batch(self, queries, opts=None):
        return self._staged("search_batch", ("limit",), queries, opts)

    def quantized_search_batch(self, queries, opts=None):
        return self._staged("quantized_search_batch", ("limit", "candidates"), queries, opts)

    def funnel_search_batch(self, queries, opts=None):
        return self._staged("funnel_search_batch", ("limit", "candidates", "stages", "dimensions"), queries, opts)
