// vt_hybrid.h -- hybrid_search on one shard (collection.ex:325-345) and the stateless helpers' shared pieces
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// ---- hybrid_search as ONE device chain --------------------------------------------------------
// collection.ex:325-345 when every generator keeps at most kMaxFusedK candidates: each generator
// ends in a device block (funnel stages chained like funnel_ready's, K4h / K4 for the quantized
// one, K1 for the index's own search), launch_union_rows merges the blocks' rows (every row once:
// hybrid_candidates, collection.ex:515-532), and the exact rerank gathers by that list, whose
// length only the device knows (K1's batch mode with one query / the f64 cosine rerank with
// n_dev).  One wait at the end instead of one per generator and stage.  The candidate SET and the
// rerank's arithmetic are the host-composed path's, so are the hits, bit for bit
// (tests/test_gpu_parity.py::test_hybrid_device_chain_equals_host_composition).
bool hybrid_fits_device(const Shard *ix, const int *kinds, const size_t *candidates, const size_t *stage_off,
                        const size_t *stages, size_t ngen, size_t limit) {
  // Opt-in (VT_HYBRID_CHAIN=1): measured, the chain is no faster than the host-composed path -- 0.843
  // vs 0.826 ms per call at N = 1 M, 0.351 vs 0.321 at 100 k, 5.745 vs 5.719 at 10 M (cosine, d = 768,
  // funnel + quantized + search generators of 100 candidates, tools/hybrid_probe.py): the generators'
  // waits cost ~15 us each against 0.7 ms of scans, and the chain pays them back in a union kernel
  // and a rerank sized for the worst case.  profiles/r03/hybrid_probe.jsonl.
  if (!vt::env::on(vt::env::HYBRID_CHAIN) || ngen == 0 || ngen > 8 || limit > (size_t)vt::kMaxFusedK) return false;
  if (ix->metric == VT_JACCARD && ix->dim >= 4096) return false;
  const uint32_t d = (uint32_t)ix->dim;
  for (size_t i = 0; i < ngen; ++i) {
    const size_t cand = std::min<size_t>(candidates[i], ix->n);
    if (cand == 0 || cand > (size_t)vt::kMaxFusedK) return false;
    if (kinds[i] == VT_GEN_FUNNEL) {
      if (!funnel_fits_device(ix, stages + stage_off[i], stage_off[i + 1] - stage_off[i], cand, cand)) return false;
    } else if (kinds[i] == VT_GEN_SEARCH) {
      if (vt::scan_lds_bytes(d, (uint32_t)cand) == 0) return false;
    }
  }
  const uint32_t k = (uint32_t)std::min<size_t>(limit, ngen * (size_t)vt::kMaxFusedK);
  if (ix->metric == VT_COSINE) return (size_t)2 * ((d + 3) / 4 * 4) * 4 + 64 <= 160 * 1024;
  return vt::scan_lds_bytes(d, k) != 0;
}

// The query is already in c.dQ (and its sign bits in c.dQbits).  kRetryInternal: the device list of
// a quantized generator overflowed on ties -- the caller takes the host-composed path.
int hybrid_dev(Shard *ix, Ctx &c, const float *query, const int *kinds, const size_t *candidates, const size_t *stage_off,
               const size_t *stages, size_t ngen, size_t limit, uint32_t qnz_full, vt_hits **out) {
  const uint32_t d = (uint32_t)ix->dim;
  uint32_t cap = 0;  // the union holds at most every generator's candidates
  for (size_t g = 0; g < ngen; ++g) cap += (uint32_t)std::min<size_t>(candidates[g], ix->n);
  const bool trace = vt::env::on(vt::env::TRACE_HYBRID);  // phases of a chain on stderr
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
  VT_TRY(c.dStageB.ensure(ngen));
  VT_TRY(c.dStage.ensure(2));
  VT_TRY(c.dRows.ensure(cap));
  VT_TRY(c.dBCount.ensure(1));
  // (the stages' own timing uses one pair of events per context: several timed stages in one chain
  // would overwrite each other's records, so a chain is counted -- hybrid_device_chains -- not timed)
  struct NoStageTiming {
    Ctx &c;
    bool was;
    explicit NoStageTiming(Ctx &c_) : c(c_), was(c_.profiling) { c.profiling = false; }
    ~NoStageTiming() { c.profiling = was; }
  } no_stage_timing(c);
  bool timed_hamming = false;
  for (size_t g = 0; g < ngen; ++g) {
    const uint32_t cand = (uint32_t)std::min<size_t>(candidates[g], ix->n);
    ResultBlock *dst = c.dStageB.p + g;
    if (kinds[g] == VT_GEN_FUNNEL) {
      const size_t *st = stages + stage_off[g];
      const size_t nst = stage_off[g + 1] - stage_off[g];
      const ResultBlock *src = nullptr;
      uint32_t count = ix->n;
      for (size_t i = 0; i < nst; ++i) {
        uint32_t nz = 0;
        for (size_t j = 0; j < st[i]; ++j) nz += query[j] != 0.0f ? 1u : 0u;
        const uint32_t want = std::min<uint32_t>(cand, count);
        ResultBlock *to = i + 1 == nst ? dst : c.dStage.p + (i & 1);
        VT_TRY(funnel_stage_dev(ix, c, query, (uint32_t)st[i], src, count, want, nz, to, false));
        src = to;
        count = want;
      }
    } else if (kinds[g] == VT_GEN_QUANTIZED) {
      const bool hist = d <= vt::kHammingHistMaxDim && ix->n >= 16384 && !vt::env::on(vt::env::HAMMING_LISTS);
      VT_TRY(hamming_stage_dev(ix, c, cand, hist, dst, &timed_hamming));
    } else {
      VT_TRY(scan_stage_dev(ix, c, d, nullptr, ix->n, cand, qnz_full, dst, false));
    }
  }
  VT_HIP(vt::launch_union_rows(c.dStageB.p, (uint32_t)ngen, c.dRows.p, c.dBCount.p, c.stream));
  // hybrid_rerank :exact == exact_rerank (collection.ex:627-630, :821-851) over the union
  const uint32_t k = (uint32_t)std::min<size_t>(limit, cap);
  if (ix->metric == VT_COSINE) {
    VT_TRY(c.dCandKeys.ensure(cap));
    VT_TRY(c.dCandPay.ensure(cap));
    vt::CosineRerankArgs a{};
    a.X = ix->dX;
    a.stride = ix->ld;
    a.q = c.qsrc;
    a.id_rank = ix->dRank.p;
    a.gather = c.dRows.p;
    a.gather_stride = 1;
    a.n = cap;
    a.n_dev = c.dBCount.p;
    a.d = d;
    a.out_keys = c.dCandKeys.p;
    a.out_pay = c.dCandPay.p;
    a.status = c.dStatus.p;
    VT_HIP(vt::launch_cosine_rerank(a, c.stream));
    // (the keys of the union's rows sit at the front of the list: the select reads only those)
    VT_HIP(vt::launch_select(c.dCandKeys.p, c.dCandPay.p, cap, k, 0, 0, c.dStatus.p, c.dResMapped, c.dSelKeys.p, c.dSelPay.p,
                             c.stream, c.dBCount.p));
  } else {
    constexpr uint32_t kBlocks = 8;
    VT_TRY(c.dPartKeys.ensure((size_t)kBlocks * k));
    VT_TRY(c.dPartPay.ensure((size_t)kBlocks * k));
    vt::ScanArgs sa{};
    sa.X = ix->dX;
    sa.stride = ix->ld;
    sa.q = c.qsrc;
    sa.id_rank = ix->dRank.p;
    sa.gather = c.dRows.p;
    sa.gather_stride = 1;
    sa.n = cap;
    sa.d = d;
    sa.metric = ix->metric;
    sa.order = ix->order;
    sa.k = k;
    sa.q_nonzero = qnz_full;
    sa.part_keys = c.dPartKeys.p;
    sa.part_pay = c.dPartPay.p;
    sa.status = c.dStatus.p;
    sa.batch_counts = c.dBCount.p;
    sa.batch_cap = cap;
    VT_HIP(vt::launch_scan_batch(sa, kBlocks, 1, c.stream));
    VT_HIP(vt::launch_select(c.dPartKeys.p, c.dPartPay.p, kBlocks * k, k, 0, 0, c.dStatus.p, c.dResMapped, c.dSelKeys.p,
                             c.dSelPay.p, c.stream));
  }
  const double t_queued = since();
  VT_HIP(hipStreamSynchronize(c.stream));
  if (trace) std::fprintf(stderr, "[vt] hybrid chain: queued %.3f ms, device done %.3f\n", t_queued, since());
  VT_TRY(c.settle_prefix_profile());
  (void)timed_hamming;  // (stage timing is off inside a chain)
  if (c.hRes.p->status == VT_ERR_OVERFLOW) return VT_ERR_OVERFLOW;
  if (c.hRes.p->status != 0) return kRetryInternal;
  c.prof.hybrid_device_chains += 1;
  std::vector<vt::Entry> entries(c.hRes.p->e, c.hRes.p->e + c.hRes.p->count);
  return make_hits(ix, entries, out);
}

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
  if (!local && hybrid_fits_device(ix, kinds, candidates, stage_off, stages, ngen, limit)) {
    const int rc = hybrid_dev(ix, c, query, kinds, candidates, stage_off, stages, ngen, limit, qnz_full, out);
    if (rc != kRetryInternal) return rc;
  }
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
