// vt_hybrid.h -- hybrid_search on one shard (collection.ex:325-345) and the stateless helpers' shared pieces
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// (r03-r05 carried hybrid_search as ONE device chain as well -- every generator ending in a device block, a union kernel,
// the rerank gathering by a list whose length only the device knew, one wait at the end; opt-in, VT_HYBRID_CHAIN=1, because
// measured it was no faster than the host-composed path below: 0.843 vs 0.826 ms per call at N = 1 M, 0.351 vs 0.321 at
// 100 k, 5.745 vs 5.719 at 10 M, profiles/r03/hybrid_probe.jsonl, DESIGN_APPENDIX A.12.  It left the library in r06.)

// collection.ex:325-345 on a shard whose ranks are strictly current (and whose sign bits
// are, when a quantized generator takes part).
int hybrid_ready(Shard *ix, Ctx &c, const float *query, size_t n, const int *kinds, const size_t *candidates,
                 const size_t *stage_off, const size_t *stages, size_t ngen, size_t limit, vt_hits **out,
                 LocalStages *local = nullptr) {
  VT_TRY(validate_vector(query, n, ix->dim));
  if (ngen == 0) return VT_ERR_ARGUMENT;
  for (size_t i = 0; i < ngen; ++i) {
    if (kinds[i] < VT_GEN_FUNNEL || kinds[i] > VT_GEN_SEARCH || candidates[i] == 0) return VT_ERR_ARGUMENT;
    if (kinds[i] == VT_GEN_FUNNEL) {
      if (stage_off[i + 1] <= stage_off[i]) return VT_ERR_PREFIX;
      for (size_t j = stage_off[i]; j < stage_off[i + 1]; ++j)
        if (stages[j] == 0 || stages[j] > n) return VT_ERR_PREFIX;
    }
  }
  if (ix->n == 0 || limit == 0) return empty_hits(out);
  uint32_t qnz_full = 0;
  VT_TRY(upload_query(c, query, n, &qnz_full, true));
  // hybrid_candidates (collection.ex:515-532): every generator's candidates, first occurrence wins
  std::vector<uint32_t> all, rows;
  std::unordered_set<uint32_t> seen;  // (a few hundred rows: never a column over the corpus)
  std::vector<vt::Entry> kept;
  if (local) local->gens.assign(ngen, {});
  for (size_t i = 0; i < ngen; ++i) {
    kept.clear();
    if (kinds[i] == VT_GEN_FUNNEL) {
      VT_TRY(funnel_rows(ix, c, query, stages + stage_off[i], stage_off[i + 1] - stage_off[i], candidates[i], rows,
                         local ? &kept : nullptr));
    } else if (kinds[i] == VT_GEN_QUANTIZED) {
      VT_TRY(quantized_rows(ix, c, candidates[i], rows, local ? &kept : nullptr));
    } else {  // the index's own search with limit = candidates (collection.ex:583-592)
      // (flat search ranks cosine by the f32 dot of normalised vectors, not by the f64 cosine
      // a vector_top_k stage would use: the plain scan serves every metric here)
      ScanJob j{};
      j.X = ix->dX;
      j.stride = ix->ld;
      j.id_rank = ix->dRank.p;
      j.n = ix->n;
      j.d = (uint32_t)ix->dim;
      j.metric = ix->metric;
      j.order = ix->order;
      j.q_nonzero = qnz_full;
      VT_TRY(run_scan(c, j, candidates[i], kept, false));
      rows.resize(kept.size());
      for (size_t r = 0; r < kept.size(); ++r) rows[r] = kept[r].row;
    }
    if (local) local->gens[i] = kept;
    for (uint32_t r : rows)
      if (seen.insert(r).second) all.push_back(r);
  }
  std::vector<vt::Entry> entries;
  if (local) {
    if (!all.empty()) VT_TRY(funnel_stage(ix, c, query, (uint32_t)ix->dim, all, false, all.size(), qnz_full, entries));
    local->final_ = std::move(entries);
    return VT_OK;
  }
  if (all.empty()) return empty_hits(out);
  // hybrid_rerank :exact == exact_rerank (collection.ex:627-630, :821-851)
  VT_TRY(funnel_stage(ix, c, query, (uint32_t)ix->dim, all, false, limit, qnz_full, entries));
  return make_hits(ix, entries, out);
}


// Per-device context for the stateless helpers.
std::mutex g_ctx_mu;
std::unordered_map<int, std::unique_ptr<Ctx>> g_ctx;
int stateless_ctx(int device, Ctx **out) {
  std::lock_guard<std::mutex> g(g_ctx_mu);
  auto it = g_ctx.find(device);
  if (it == g_ctx.end()) {
    auto c = std::make_unique<Ctx>();
    VT_TRY(c->init(device));
    it = g_ctx.emplace(device, std::move(c)).first;
  }
  *out = it->second.get();
  return (*out)->bind();
}

// id_rank for an ad-hoc batch of ids (ties between equal ids: input order).
void ranks_for_ids(const char *ids, const size_t *id_off, size_t count, std::vector<uint32_t> &rank) {
  std::vector<uint32_t> order(count);
  for (size_t i = 0; i < count; ++i) order[i] = (uint32_t)i;
  auto view = [&](uint32_t i) { return std::pair<const char *, size_t>(ids + id_off[i], id_off[i + 1] - id_off[i]); };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    auto x = view(a), y = view(b);
    const size_t m = std::min(x.second, y.second);
    const int c = m ? std::memcmp(x.first, y.first, m) : 0;
    if (c) return c < 0;
    return x.second < y.second;
  });
  rank.resize(count);
  for (size_t i = 0; i < count; ++i) rank[order[i]] = (uint32_t)i;
}

int hits_from_batch(const char *ids, const size_t *id_off, const std::vector<vt::Entry> &entries, vt_hits **out) {
  auto h = std::make_unique<vt_hits>();
  for (const auto &e : entries) {
    h->ids.emplace_back(ids + id_off[e.row], id_off[e.row + 1] - id_off[e.row]);
    h->raw.push_back(e.raw);
    h->rank_key.push_back(rank_key_of(e.key));
  }
  *out = h.release();
  return VT_OK;
}

}  // namespace
