// vt_search.h -- what a reader needs current, flat search on a shard, the f64 cosine scan, funnel stages, sign bits, norms.
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// ---- what a reader needs up to date before it may run under the shared lock ----------
enum : unsigned { NEED_RANKS = 1, NEED_STRICT_RANKS = 2, NEED_BITS = 4, NEED_NORMS = 8, NEED_NZBITS = 16 };
// Internal: only the true id order can decide (a tie at the boundary of a lazy search).
constexpr int kEscalate = -101;

// Ids inserted out of order since the last ranking keep one shared sentinel rank; a search
// that wants `limit` hits may run on that column if it can ask for one hit more (see
// search_ready).  Past ~1/8 of the rows unranked the eventual rebuild would have to sort
// too many ids at once: rebuild now, while it is still cheap.
bool lazy_ranks_ok(const Shard *ix, size_t limit) {
  const size_t lazy_want = std::min<size_t>(limit, ix->n) + (limit < ix->n ? 1 : 0);
  // (one select pass only: very wide rows leave LDS for the small candidate buffer alone)
  const size_t kmax = vt::scan_lds_bytes((uint32_t)ix->dim, vt::kMaxFusedK) ? (size_t)vt::kMaxFusedK : (size_t)vt::kSmallK;
  return !ix->ranks_clean && !ix->external_ranks && lazy_want <= kmax &&
         ix->unranked <= std::max<size_t>(65536, ix->n / 8);
}

// Float hamming and jaccard compare which coordinates are non-zero and nothing else
// (distances.rs:319-347), so on a collection under one of them flat_search is K4 over a column of
// non-zero bits -- d / 8 bytes per row instead of 4 d, the same integers, the same f32 score, the
// same keys -- once the corpus is big enough for the column to pay for its upkeep and as long as
// the hits fit a few K4 passes (K1's one-pass threshold path serves the larger limits).
// NEED_NZBITS asks for that column and means nothing where this says no.
constexpr size_t kPatternMinRows = 16384, kPatternMaxWant = 8 * (size_t)vt::kMaxFusedK;
bool pattern_metric(int metric) {
  return metric == VT_HAMMING || metric == VT_JACCARD;
}
bool pattern_search_applies(const Shard *ix, size_t limit) {
  return pattern_metric(ix->metric) && !ix->nz_refused && ix->n >= kPatternMinRows &&
         std::min<size_t>(limit, ix->n) + 1 <= kPatternMaxWant;
}

// The bf16 shadow of the rows (K2s, vt_batch_shadow.hip) travels with NEED_NORMS: both are what a
// matrix-core batch wants current, and nothing else reads either.
bool shadow_metric(int metric) {
  return metric == VT_COSINE || metric == VT_INNER_PRODUCT || metric == VT_NEG_INNER_PRODUCT || metric == VT_L2 ||
         metric == VT_L2_SQUARED;
}
bool shadow_wanted(const Shard *ix) {
  return ix->shadow_mode == VT_SHADOW_AUTO && !ix->sh_refused && ix->nominate == VT_NOMINATE_BF16 && shadow_metric(ix->metric) &&
         ix->n > 0 && ix->dim > 0;
}
size_t shadow_rows(const Shard *ix) { return std::max<size_t>(ix->cap, ix->n); }
bool shadow_current(const Shard *ix) {
  return ix->sh_valid && ix->sh_dirty.empty() && ix->dShadow.p != nullptr &&
         ix->dShadow.count >= vt::shadow_elems((uint32_t)shadow_rows(ix), ix->ld);
}

bool shard_stale(const Shard *ix, unsigned need, size_t limit) {
  if (ix->n == 0) return false;
  const size_t rows = std::max<size_t>(ix->cap, ix->n);
  if ((need & (NEED_RANKS | NEED_STRICT_RANKS)) && !ix->ranks_clean) {
    if ((need & NEED_STRICT_RANKS) || !lazy_ranks_ok(ix, limit)) return true;
    if (!ix->rank_dirty.empty() || ix->rank_dirty_all || ix->dRank.count < rows) return true;
  }
  if (need & NEED_BITS) {
    const size_t bwords = vt::hamming_matrix_words((uint32_t)rows, ((uint32_t)ix->dim + 63) / 64);
    if (!ix->bits_valid || !ix->bits_dirty.empty() || ix->dBits.count < bwords) return true;
  }
  if ((need & NEED_NZBITS) && pattern_search_applies(ix, limit)) {
    const size_t bwords = vt::hamming_matrix_words((uint32_t)rows, ((uint32_t)ix->dim + 63) / 64);
    if (!ix->nz_valid || !ix->nz_dirty.empty() || ix->dNzBits.count < bwords) return true;
  }
  if ((need & NEED_NORMS) && (ix->max_sqnorm < 0.0 || !ix->norm_dirty.empty() || ix->dXnorm2.count < rows)) return true;
  if ((need & NEED_NORMS) && shadow_wanted(ix) && !shadow_current(ix)) return true;
  return false;
}

int index_ensure_bits(Shard *ix, bool nonzero = false);
int index_ensure_norms(Shard *ix);
int index_ensure_shadow(Shard *ix);

// Brings the derived columns a reader needs up to date (exclusive access; primary context).
int shard_prepare(Shard *ix, unsigned need, size_t limit) {
  if (ix->n == 0) return VT_OK;
  if (need & (NEED_RANKS | NEED_STRICT_RANKS)) {
    if (!(need & NEED_STRICT_RANKS) && lazy_ranks_ok(ix, limit)) VT_TRY(index_lazy_ranks(ix));
    else VT_TRY(index_sync_ranks(ix, false));
  }
  if (need & NEED_BITS) VT_TRY(index_ensure_bits(ix));
  if ((need & NEED_NZBITS) && pattern_search_applies(ix, limit)) VT_TRY(index_ensure_bits(ix, true));
  if (need & NEED_NORMS) VT_TRY(index_ensure_norms(ix));
  if ((need & NEED_NORMS) && shadow_wanted(ix)) VT_TRY(index_ensure_shadow(ix));
  return VT_OK;
}

int batch_group(Shard *ix, Ctx &c, const float *queries, size_t nq, size_t limit, vt_hits **out, std::vector<char> &done,
                bool bf16, const float *tau_given, std::vector<float> *retry_tau);

// A lone search as a batch of one through the bf16 shadow (vt_flat_set_single_nominate): applies when everything the
// batch path wants is current already -- a reader under the shared lock never waits for a build.
bool single_nominate_applies(const Shard *ix, size_t limit) {
  return ix->single_nominate && ix->ranks_clean && limit >= 1 && limit <= (size_t)vt::kMaxFusedK && ix->n >= 4096 &&
         shadow_wanted(ix) && shadow_current(ix) && !shard_stale(ix, NEED_NORMS, limit);
}

// flat.rs:96-124 on a shard whose rank column shard_prepare has brought up to date --
// strictly (ranks_clean) or lazily (newcomers share kUnranked).  Read-only on the shard.
int search_ready(Shard *ix, Ctx &c, const float *query, size_t n, size_t limit, vt_hits **out) {
  if (limit == 0) return empty_hits(out);
  VT_TRY(validate_vector(query, n, ix->dim));
  if (ix->n == 0) return empty_hits(out);
  if (single_nominate_applies(ix, limit)) {
    std::vector<char> done(1, 0);
    vt_hits *one = nullptr;
    VT_TRY(batch_group(ix, c, query, 1, limit, &one, done, true, nullptr, nullptr));
    if (done[0]) {
      *out = one;
      return VT_OK;
    }
    delete one;  // (not certified: the exact scan decides)
    c.prof.batch_fallbacks += 1;
  }
  const bool lazy = !ix->ranks_clean;
  const size_t lazy_want = std::min<size_t>(limit, ix->n) + (limit < ix->n ? 1 : 0);
  // (a non-zero-bit column that is not current is simply not used: the rows always are)
  const bool by_pattern = pattern_search_applies(ix, limit) && !shard_stale(ix, NEED_NZBITS, limit);
  uint32_t qnz = 0;
  VT_TRY(upload_query(c, query, n, &qnz, by_pattern ? 2 : 0));
  ScanJob j{};
  j.X = ix->dX;
  j.stride = ix->ld;
  j.id_rank = ix->dRank.p;
  j.gather = nullptr;
  j.gather_stride = 0;
  j.n = ix->n;
  j.d = (uint32_t)ix->dim;
  j.metric = ix->metric;
  j.order = ix->order;
  j.q_nonzero = qnz;
  std::vector<vt::Entry> entries;
  auto best_rows = [&](size_t want) -> int {
    if (by_pattern)
      return run_hamming(c, ix->dNzBits.p, c.dQbits, ix->dRank.p, ix->n, j.d, want, entries, true, ix->metric == VT_JACCARD);
    return run_scan(c, j, want, entries, true);
  };
  if (lazy) {
    // one hit more than asked for: if it does not tie with the last wanted one, the set is
    // exact whatever the unranked rows' id order is, and equal-rank runs are put in id order here
    VT_TRY(best_rows(lazy_want));
    const bool ambiguous = limit < ix->n && entries.size() == lazy_want &&
                           rank_key_of(entries[limit - 1].key) == rank_key_of(entries[limit].key);
    if (ambiguous) return kEscalate;  // a tie across the boundary: only the true id order can cut it
    if (entries.size() > limit) entries.resize(limit);
    for (size_t i = 0; i < entries.size();) {
      size_t e = i + 1;
      while (e < entries.size() && rank_key_of(entries[e].key) == rank_key_of(entries[i].key)) ++e;
      if (e - i > 1)
        std::sort(entries.begin() + i, entries.begin() + e,
                  [&](const vt::Entry &a, const vt::Entry &b) { return ix->ids[a.row] < ix->ids[b.row]; });
      i = e;
    }
    return make_hits(ix, entries, out);
  }
  VT_TRY(best_rows(limit));
  return make_hits(ix, entries, out);
}

// The same for a caller that owns the shard outright (a shard worker, or any caller under
// the exclusive lock): prepare, run on the primary context, settle a boundary tie.
int search_owner(Shard *ix, const float *query, size_t n, size_t limit, vt_hits **out) {
  if (shard_stale(ix, NEED_RANKS | NEED_NZBITS, limit)) VT_TRY(shard_prepare(ix, NEED_RANKS | NEED_NZBITS, limit));
  int st = search_ready(ix, ix->ctx, query, n, limit, out);
  if (st == kEscalate) {
    VT_TRY(shard_prepare(ix, NEED_STRICT_RANKS, limit));
    st = search_ready(ix, ix->ctx, query, n, limit, out);
  }
  return st;
}

// Exact f64-cosine scan of the first `d` coordinates of every row (K6b), passes
// of <= kMaxFusedK until `want` hits are collected.  Query already in c.dQ.
int run_cosine_scan(Ctx &c, Shard *ix, uint32_t d, double qq, size_t want, std::vector<vt::Entry> &out) {
  const size_t kmax = vt::cosine_scan_lds_bytes(d, vt::kMaxFusedK) ? (size_t)vt::kMaxFusedK : (size_t)vt::kSmallK;
  if (vt::cosine_scan_lds_bytes(d, 1) == 0) return fail(VT_ERR_UNSUPPORTED, "prefix too long for the cosine scan kernel");
  uint64_t lo = 0;
  bool has_lo = false;
  const size_t total = std::min<size_t>(want, ix->n);
  // (a prefix-cosine pass costs ~90 us before its first byte: f64 sums, its own select and wait)
  if (out.empty() && threshold_applies(total, ix->n, (double)ix->n * vt::padded_dim(d) * 4.0, 90e-6)) {
    const uint32_t k = (uint32_t)total;
    VT_TRY(c.dKeyCol.ensure(((size_t)ix->n + 1) / 2 * 2));
    VT_TRY(c.dPartKeys.ensure(kThresholdListCap));
    VT_TRY(c.dPartPay.ensure(kThresholdListCap));
    vt::CosineScanArgs a{};
    a.X = ix->dX;
    a.stride = ix->ld;
    a.q = c.qsrc;
    a.qq = qq;
    a.id_rank = ix->dRank.p;
    a.n = ix->n;
    a.d = d;
    a.k = 1;
    a.part_keys = c.dPartKeys.p;
    a.part_pay = c.dPartPay.p;
    a.status = c.dStatus.p;
    a.key_out = c.dKeyCol.p;
    if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
    VT_HIP(vt::launch_cosine_scan(a, c.grid_for((ix->n + 63) / 64, vt::cosine_scan_lds_bytes(d, 1)), c.stream));
    if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
    VT_TRY(threshold_rows(c, ix->n, k));
    VT_TRY(c.dCandKeys.ensure(k));
    VT_TRY(c.dCandPay.ensure(k));
    vt::CosineRerankArgs g{};
    g.X = ix->dX;
    g.stride = ix->ld;
    g.q = c.qsrc;
    g.id_rank = ix->dRank.p;
    g.gather = &c.dListPay.p->row;
    g.gather_stride = sizeof(vt::Payload) / sizeof(uint32_t);
    g.n = k;
    g.d = d;
    g.out_keys = c.dCandKeys.p;
    g.out_pay = c.dCandPay.p;
    g.status = c.dStatus.p;
    VT_HIP(vt::launch_cosine_rerank(g, c.stream));
    const int rc = collect_sorted_list(c, c.dCandKeys.p, c.dCandPay.p, k, out);
    if (c.profiling && rc != kRetryInternal) {
      float ms = 0.0f;
      VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
      c.prof.prefix_launches += 1;
      c.prof.prefix_ms += ms;
      c.prof.prefix_bytes += (uint64_t)ix->n * d * 4;
    }
    if (rc != kRetryInternal) return rc;
    out.clear();
  }
  while (out.size() < total) {
    const uint32_t k = (uint32_t)std::min<size_t>(kmax, total - out.size());
    const uint32_t blocks = c.grid_for((ix->n + 63) / 64, vt::cosine_scan_lds_bytes(d, k));
    VT_TRY(c.dPartKeys.ensure((size_t)blocks * k));
    VT_TRY(c.dPartPay.ensure((size_t)blocks * k));
    vt::CosineScanArgs a{};
    a.X = ix->dX;
    a.stride = ix->ld;
    a.q = c.qsrc;
    a.qq = qq;
    a.id_rank = ix->dRank.p;
    a.n = ix->n;
    a.d = d;
    a.k = k;
    a.lo_key = lo;
    a.has_lo = has_lo ? 1 : 0;
    a.part_keys = c.dPartKeys.p;
    a.part_pay = c.dPartPay.p;
    a.status = c.dStatus.p;
    if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
    VT_HIP(vt::launch_cosine_scan(a, blocks, c.stream));
    if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
    VT_TRY(select_pass(c, c.dPartKeys.p, c.dPartPay.p, blocks * k, k, 0, false));
    if (c.profiling) {
      float ms = 0.0f;
      VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
      c.prof.prefix_launches += 1;
      c.prof.prefix_ms += ms;
      c.prof.prefix_bytes += (uint64_t)ix->n * d * 4;
    }
    if (c.hRes.p->status == VT_ERR_OVERFLOW) return VT_ERR_OVERFLOW;
    const uint32_t got = c.hRes.p->count;
    for (uint32_t i = 0; i < got; ++i) out.push_back(c.hRes.p->e[i]);
    if (got < k) break;
    lo = c.hRes.p->e[got - 1].key;
    has_lo = true;
  }
  return VT_OK;
}

// A stage over ALL rows of a float hamming / jaccard collection -- vector_top_k on the first d coordinates
// (search.rs:38-73; distances.rs:319-347 look at nothing but which coordinates are non-zero) -- reads the
// first ceil(d / 64) words of every row's non-zero bits instead of the rows: K4 with a prefix mask, 1/32 of the
// bytes.  Applies when the column is current (a reader never waits for it) and the query's non-zero bits are
// what c.dQbits holds (upload_query with_bits = 2).
bool pattern_stage_applies(const Shard *ix, const Ctx &c, size_t want) {
  return c.qbits_kind == 2 && want >= 1 && want <= (size_t)vt::kMaxFusedK && pattern_search_applies(ix, want) &&
         !shard_stale(ix, NEED_NZBITS, want);
}

// One vector_top_k stage (search.rs:38-73) on the resident corpus: prefix length
// `d`, over all rows (`rows` empty) or over the candidate rows of the previous
// stage; keeps `want` hits.
int funnel_stage(Shard *ix, Ctx &c, const float *query, uint32_t d, const std::vector<uint32_t> &rows, bool all_rows,
                 size_t want, uint32_t qnz, std::vector<vt::Entry> &out) {
  if (!all_rows) {
    VT_TRY(c.dRows.ensure(rows.size()));
    VT_HIP(hipMemcpyAsync(c.dRows.p, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
    VT_HIP(hipStreamSynchronize(c.stream));
  }
  if (ix->metric == VT_COSINE) {
    double qq = 0.0;  // f64_dot(q, q) over the prefix (distances.rs:179-185)
    for (uint32_t j = 0; j < d; ++j) qq += (double)query[j] * (double)query[j];
    if (all_rows) return run_cosine_scan(c, ix, d, qq, want, out);
    VT_TRY(c.dCandKeys.ensure(rows.size()));
    VT_TRY(c.dCandPay.ensure(rows.size()));
    vt::CosineRerankArgs a{};
    a.X = ix->dX;
    a.stride = ix->ld;
    a.q = c.qsrc;
    a.id_rank = ix->dRank.p;
    a.gather = c.dRows.p;
    a.gather_stride = 1;
    a.n = (uint32_t)rows.size();
    a.d = d;
    a.out_keys = c.dCandKeys.p;
    a.out_pay = c.dCandPay.p;
    a.status = c.dStatus.p;
    VT_HIP(vt::launch_cosine_rerank(a, c.stream));
    return collect_from_keys(c, c.dCandKeys.p, c.dCandPay.p, (uint32_t)rows.size(), want, out);
  }
  if (all_rows && pattern_stage_applies(ix, c, want))
    return run_hamming(c, ix->dNzBits.p, c.dQbits, ix->dRank.p, ix->n, d, want, out, true, ix->metric == VT_JACCARD,
                       (((uint32_t)ix->dim + 63) / 64 + 1) / 2);
  ScanJob j{};
  j.X = ix->dX;
  j.stride = ix->ld;
  j.id_rank = ix->dRank.p;
  j.gather = all_rows ? nullptr : c.dRows.p;
  j.gather_stride = 1;
  j.n = all_rows ? ix->n : (uint32_t)rows.size();
  j.d = d;
  j.metric = ix->metric;
  j.order = ix->order;
  j.q_nonzero = qnz;
  return run_scan(c, j, want, out, false);
}

// The index's own metric (K1 arithmetic: for a cosine index the f32 dot of normalised rows, like
// flat_search) over the first `d` coordinates of `count` rows -- all rows, or the Entry.row column
// of `src` -- on the device: the `want` <= kMaxFusedK best, sorted, into `dst`.  Nothing is waited
// for; an overflow flag stays in c.dStatus until a select with `last` moves it into its block.
int scan_stage_dev(Shard *ix, Ctx &c, uint32_t d, const ResultBlock *src, uint32_t count, uint32_t want, uint32_t qnz,
                   ResultBlock *dst, bool last) {
  const uint32_t *gather = src ? &src->e[0].row : nullptr;
  const uint32_t gstride = sizeof(vt::Entry) / sizeof(uint32_t);
  int *status = last ? c.dStatus.p : nullptr;
  if (!src && pattern_stage_applies(ix, c, want)) {
    // all rows under float hamming / jaccard: the non-zero bits of the prefix (K4) instead of the rows
    const uint32_t words = (d + 63) / 64;
    const uint32_t hblocks = c.grid_for((ix->n + 63) / 64, vt::hamming_lds_bytes(want), c.hamming_blocks_per_cu);
    const uint32_t hlists = vt::scan_lists(hblocks);
    VT_TRY(c.dPartKeys.ensure((size_t)hlists * want));
    VT_TRY(c.dPartPay.ensure((size_t)hlists * want));
    vt::HammingArgs h{};
    h.bits = ix->dNzBits.p;
    h.qbits = c.dQbits;
    h.id_rank = ix->dRank.p;
    h.n = ix->n;
    h.words = words;
    h.pairs = (words + 1) / 2;
    h.d = d;
    h.k = want;
    h.part_keys = c.dPartKeys.p;
    h.part_pay = c.dPartPay.p;
    h.jaccard = ix->metric == VT_JACCARD ? 1 : 0;
    h.tile_pairs = (((uint32_t)ix->dim + 63) / 64 + 1) / 2;
    if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
    VT_HIP(vt::launch_hamming(h, hblocks, c.stream));
    if (c.profiling) {
      VT_HIP(hipEventRecord(c.ev1, c.stream));
      c.prefix_pending += 1;
      c.prof.prefix_bytes += (uint64_t)ix->n * words * 8;
    }
    VT_HIP(vt::launch_select(c.dPartKeys.p, c.dPartPay.p, hlists * want, want, 0, 0, status, dst, c.dSelKeys.p, c.dSelPay.p, c.stream));
    return VT_OK;
  }
  const uint32_t tile_rows = vt::scan_tile_rows(count, d, c.resident_waves());
  const uint32_t ntiles = (count + tile_rows - 1) / tile_rows;
  const uint32_t blocks = c.grid_for(ntiles, vt::scan_lds_bytes(d, want));
  const uint32_t lists = vt::scan_lists(blocks);
  VT_TRY(c.dPartKeys.ensure((size_t)lists * want));
  VT_TRY(c.dPartPay.ensure((size_t)lists * want));
  vt::ScanArgs a{};
  a.X = ix->dX;
  a.stride = ix->ld;
  a.q = c.qsrc;
  a.id_rank = ix->dRank.p;
  a.gather = gather;
  a.gather_stride = gather ? gstride : 0;
  a.n = count;
  a.d = d;
  a.metric = ix->metric;
  a.order = ix->order;
  a.k = want;
  a.q_nonzero = qnz;
  a.tile_rows = tile_rows;
  a.part_keys = c.dPartKeys.p;
  a.part_pay = c.dPartPay.p;
  a.status = c.dStatus.p;
  const bool timed = c.profiling && !src && d < (uint32_t)ix->dim;  // a sweep of every row's prefix: priced like the cosine one
  if (timed) VT_HIP(hipEventRecord(c.ev0, c.stream));
  VT_HIP(vt::launch_scan(a, blocks, c.stream));
  if (timed) {
    VT_HIP(hipEventRecord(c.ev1, c.stream));
    c.prefix_pending += 1;
    c.prof.prefix_bytes += (uint64_t)ix->n * d * 4;
  }
  VT_HIP(vt::launch_select(c.dPartKeys.p, c.dPartPay.p, lists * want, want, 0, 0, status, dst, c.dSelKeys.p, c.dSelPay.p,
                           c.stream));
  return VT_OK;
}

// One funnel / rerank stage that never leaves the device: scores `count` rows
// (all rows, or the Entry.row column of the previous stage's block), keeps
// `want` <= kMaxFusedK of them in `dst`.  Nothing is waited for; an overflow
// flag raised by any stage stays in c.dStatus until a select with `last` moves
// it into its block.
int funnel_stage_dev(Shard *ix, Ctx &c, const float *query, uint32_t d, const ResultBlock *src, uint32_t count,
                     uint32_t want, uint32_t qnz, ResultBlock *dst, bool last) {
  if (ix->metric != VT_COSINE) return scan_stage_dev(ix, c, d, src, count, want, qnz, dst, last);
  const uint32_t *gather = src ? &src->e[0].row : nullptr;
  const uint32_t gstride = sizeof(vt::Entry) / sizeof(uint32_t);
  int *status = last ? c.dStatus.p : nullptr;
  {
    double qq = 0.0;  // f64_dot(q, q) over the prefix (distances.rs:179-185)
    for (uint32_t j = 0; j < d; ++j) qq += (double)query[j] * (double)query[j];
    if (!src) {
      const uint32_t blocks = c.grid_for((ix->n + 63) / 64, vt::cosine_scan_lds_bytes(d, want));
      VT_TRY(c.dPartKeys.ensure((size_t)blocks * want));
      VT_TRY(c.dPartPay.ensure((size_t)blocks * want));
      vt::CosineScanArgs a{};
      a.X = ix->dX;
      a.stride = ix->ld;
      a.q = c.qsrc;
      a.qq = qq;
      a.id_rank = ix->dRank.p;
      a.n = ix->n;
      a.d = d;
      a.k = want;
      a.part_keys = c.dPartKeys.p;
      a.part_pay = c.dPartPay.p;
      a.status = c.dStatus.p;
      if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
      VT_HIP(vt::launch_cosine_scan(a, blocks, c.stream));
      if (c.profiling) {
        VT_HIP(hipEventRecord(c.ev1, c.stream));
        c.prefix_pending += 1;
        c.prof.prefix_bytes += (uint64_t)ix->n * d * 4;
      }
      VT_HIP(vt::launch_select(c.dPartKeys.p, c.dPartPay.p, blocks * want, want, 0, 0, status, dst, c.dSelKeys.p,
                               c.dSelPay.p, c.stream));
      return VT_OK;
    }
    VT_TRY(c.dCandKeys.ensure(count));
    VT_TRY(c.dCandPay.ensure(count));
    vt::CosineRerankArgs a{};
    a.X = ix->dX;
    a.stride = ix->ld;
    a.q = c.qsrc;
    a.id_rank = ix->dRank.p;
    a.gather = gather;
    a.gather_stride = gstride;
    a.n = count;
    a.d = d;
    a.out_keys = c.dCandKeys.p;
    a.out_pay = c.dCandPay.p;
    a.status = c.dStatus.p;
    VT_HIP(vt::launch_cosine_rerank(a, c.stream));
    VT_HIP(vt::launch_select(c.dCandKeys.p, c.dCandPay.p, count, want, 0, 0, status, dst, c.dSelKeys.p, c.dSelPay.p,
                             c.stream));
    return VT_OK;
  }
}

// True when every stage of a funnel fits one fused pass on the device.
bool funnel_fits_device(const Shard *ix, const size_t *stages, size_t nstages, size_t candidates, size_t limit) {
  if (candidates > (size_t)vt::kMaxFusedK || limit > (size_t)vt::kMaxFusedK) return false;
  if (ix->metric == VT_JACCARD && ix->dim >= 4096) return false;
  auto fits = [&](uint32_t d, uint32_t k, bool all_rows) {
    if (ix->metric == VT_COSINE) return all_rows ? vt::cosine_scan_lds_bytes(d, k) != 0 : (size_t)2 * d * 4 + 64 <= 160 * 1024;
    return vt::scan_lds_bytes(d, k) != 0;
  };
  for (size_t i = 0; i < nstages; ++i)
    if (!fits((uint32_t)stages[i], (uint32_t)std::min<size_t>(candidates, ix->n), i == 0)) return false;
  return fits((uint32_t)ix->dim, (uint32_t)std::min<size_t>(limit, ix->n), false);
}

// Candidate rows of one funnel pass (collection.ex:674-691) without the final rerank.
int funnel_rows(Shard *ix, Ctx &c, const float *query, const size_t *stages, size_t nstages, size_t candidates,
                std::vector<uint32_t> &rows, std::vector<vt::Entry> *first = nullptr) {
  rows.clear();
  if (first) first->clear();
  bool all_rows = true;
  for (size_t i = 0; i < nstages; ++i) {
    uint32_t nz = 0;
    for (size_t j = 0; j < stages[i]; ++j) nz += query[j] != 0.0f ? 1u : 0u;
    std::vector<vt::Entry> kept;
    VT_TRY(funnel_stage(ix, c, query, (uint32_t)stages[i], rows, all_rows, candidates, nz, kept));
    if (first && i == 0) *first = kept;  // the only stage that cuts: later ones re-score the same set
    rows.resize(kept.size());
    for (size_t r = 0; r < kept.size(); ++r) rows[r] = kept[r].row;
    all_rows = false;
    if (rows.empty()) break;
  }
  return VT_OK;
}

// Sign bits of every stored row in K4's layout, built on first use (`nonzero`: the column of
// non-zero bits that flat_search under float hamming / jaccard reads instead of the rows).
int index_ensure_bits(Shard *ix, bool nonzero) {
  Ctx &c = ix->ctx;
  const uint32_t d = (uint32_t)ix->dim, words = (d + 63) / 64;
  DevBuf<uint64_t> &col = nonzero ? ix->dNzBits : ix->dBits;
  bool &valid = nonzero ? ix->nz_valid : ix->bits_valid;
  std::vector<uint32_t> &dirty = nonzero ? ix->nz_dirty : ix->bits_dirty;
  // compress_sign_bits of every stored row (collection.ex:926): kept in HBM
  const size_t bwords = vt::hamming_matrix_words(std::max<uint32_t>(ix->cap, ix->n), words);
  if (valid && col.count >= bwords) {
    // only the rows mutated since the last use
    uint32_t count = 0;
    VT_TRY(upload_row_list(ix, dirty, &count));
    VT_HIP(vt::launch_sign_pack_rows(ix->dX, ix->ld, c.dRankPairs.p, count, d, col.p, c.stream, nonzero ? 1 : 0));
    if (count) VT_HIP(hipStreamSynchronize(c.stream));  // the pinned list is reused by the next caller
    dirty.clear();
    return VT_OK;
  }
  dirty.clear();
  if (nonzero) {
    // (an accelerator, not a requirement: on a card too full for d / 8 more bytes per row the
    // searches keep reading the rows)
    valid = false;
    bool refused = col.ensure(bwords) != VT_OK;
#ifdef VT_TEST_HOOKS
    // (libvettore_hip_hooks.so only: the allocation "fails", tests/test_gpu_parity.py checks what follows)
    if (vt::env::on(vt::env::TEST_REFUSE_NZBITS)) {
      col.release();
      refused = true;
    }
#endif
    if (refused) {
      ix->nz_refused = true;
      return VT_OK;
    }
  } else {
    VT_TRY(col.ensure(bwords));
  }
  VT_HIP(hipMemsetAsync(col.p, 0, bwords * sizeof(uint64_t), c.stream));
  VT_HIP(vt::launch_sign_pack(ix->dX, ix->ld, ix->n, d, col.p, 1, c.stream, nonzero ? 1 : 0));
  // The column counts as current from here on, for readers on OTHER streams too (leased contexts
  // under the shared lock): the build must have finished before the exclusive lock can drop, also on
  // the paths that return without touching this stream again (limit == 0, a query that fails
  // validation) -- ADVICE r3.
  VT_HIP(hipStreamSynchronize(c.stream));
  valid = true;
  return VT_OK;
}

// The bf16 shadow of the rows brought up to date (exclusive access; primary context): the rows
// mutated since its last use are re-rounded in place; a first use, a slab that outgrew it or more
// than kMaxDerivedDirty mutations rebuild the whole image (one pass over the rows: ~10 ms per
// 30 GB).  An accelerator, not a requirement: without room for it -- after the allocation at least
// a quarter of the card must still be free, the rows will want to grow -- the shard is marked
// `sh_refused` and its batches keep streaming the f32 rows (until the index is emptied).
int index_ensure_shadow(Shard *ix) {
  Ctx &c = ix->ctx;
  const uint32_t rows = (uint32_t)shadow_rows(ix);
  const size_t elems = vt::shadow_elems(rows, ix->ld);
  if (ix->sh_valid && ix->dShadow.p && ix->dShadow.count >= elems) {
    uint32_t count = 0;
    VT_TRY(upload_row_list(ix, ix->sh_dirty, &count));
    VT_HIP(vt::launch_shadow_rows(ix->dX, ix->ld, c.dRankPairs.p, count, ix->ld, ix->dShadow.p, c.stream));
    if (count) VT_HIP(hipStreamSynchronize(c.stream));  // the pinned list is reused by the next caller; readers on other streams follow
    c.prof.shadow_patched_rows += count;
    ix->sh_dirty.clear();
    return VT_OK;
  }
  ix->sh_valid = false;
  ix->sh_dirty.clear();
  if (ix->dShadow.count < elems) {
    ix->dShadow.release();
    size_t free_b = 0, total_b = 0;
    VT_HIP(hipMemGetInfo(&free_b, &total_b));
    bool refused = free_b < elems * sizeof(uint16_t) || free_b - elems * sizeof(uint16_t) < total_b / 4;
    // (geometric head room: a corpus that arrives in appends would otherwise re-allocate at every growth of the slab)
    size_t want = elems;
    if (!refused) {
      const size_t roomy = elems + elems / 4;
      if (free_b >= roomy * sizeof(uint16_t) && free_b - roomy * sizeof(uint16_t) >= total_b / 4) want = roomy;
      refused = ix->dShadow.ensure(want) != VT_OK;
    }
#ifdef VT_TEST_HOOKS
    // (libvettore_hip_hooks.so only: the allocation "fails", tests/test_gpu_shadow.py checks what follows)
    if (vt::env::on(vt::env::TEST_REFUSE_SHADOW)) {
      ix->dShadow.release();
      refused = true;
    }
#endif
    if (refused) {
      ix->sh_refused = true;
      return VT_OK;
    }
  }
  const uint32_t rows_img = (uint32_t)(vt::shadow_elems(rows, ix->ld) / ix->ld);
  if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
  VT_HIP(vt::launch_shadow_build(ix->dX, ix->ld, std::min<uint32_t>(rows, ix->cap), rows_img, ix->ld, ix->dShadow.p, c.stream));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
  // current from here on, for readers on other streams too: the build has finished before the
  // exclusive lock can drop
  VT_HIP(hipStreamSynchronize(c.stream));
  if (c.profiling) {
    float ms = 0.f;
    VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    c.prof.shadow_builds += 1;
    c.prof.shadow_build_ms += ms;
  }
  ix->sh_valid = true;
  return VT_OK;
}

// binary_top_k candidates (search.rs:76-92) of the query already in c.dQ.
int quantized_rows(Shard *ix, Ctx &c, size_t candidates, std::vector<uint32_t> &rows, std::vector<vt::Entry> *entries = nullptr) {
  const uint32_t d = (uint32_t)ix->dim;
  std::vector<vt::Entry> local;
  std::vector<vt::Entry> &cand = entries ? *entries : local;
  cand.clear();
  VT_TRY(run_hamming(c, ix->dBits.p, c.dQbits, ix->dRank.p, ix->n, d, candidates, cand, false));
  rows.resize(cand.size());
  for (size_t i = 0; i < cand.size(); ++i) rows[i] = cand[i].row;
  return VT_OK;
}

// Per-row squared norms and their maximum (the error margin of the batched path), brought
// up to date: all rows on first use, afterwards only the rows mutated since.
int index_ensure_norms(Shard *ix) {
  Ctx &c = ix->ctx;
  const uint32_t d = (uint32_t)ix->dim, n = ix->n;
  VT_TRY(c.dBNorm.ensure(1));
  if (ix->max_sqnorm >= 0.0 && ix->dXnorm2.count >= std::max<uint32_t>(ix->cap, n) && !ix->norm_dirty.empty()) {
    // norms of the rows mutated since the last batch; the maximum can only be kept or raised
    // (a stale larger bound only widens the acceptance margin)
    uint32_t count = 0;
    VT_TRY(upload_row_list(ix, ix->norm_dirty, &count));
    unsigned long long bits = 0;
    std::memcpy(&bits, &ix->max_sqnorm, sizeof(double));
    VT_HIP(hipMemcpyAsync(c.dBNorm.p, &bits, sizeof(bits), hipMemcpyHostToDevice, c.stream));
    VT_HIP(vt::launch_row_sqnorms_rows(ix->dX, ix->ld, c.dRankPairs.p, count, d, ix->dXnorm2.p, c.dBNorm.p, c.stream));
    VT_HIP(hipMemcpyAsync(&bits, c.dBNorm.p, sizeof(bits), hipMemcpyDeviceToHost, c.stream));
    VT_HIP(hipStreamSynchronize(c.stream));
    std::memcpy(&ix->max_sqnorm, &bits, sizeof(double));
    ix->norm_dirty.clear();
  } else if (ix->max_sqnorm >= 0.0 && ix->dXnorm2.count < std::max<uint32_t>(ix->cap, n)) {
    ix->max_sqnorm = -1.0;  // the slab grew past the norm column
  }
  if (ix->max_sqnorm < 0.0) {
    ix->norm_dirty.clear();
    unsigned long long bits = 0;
    VT_TRY(ix->dXnorm2.ensure(std::max<uint32_t>(ix->cap, n)));
    VT_HIP(hipMemsetAsync(c.dBNorm.p, 0, sizeof(unsigned long long), c.stream));
    VT_HIP(vt::launch_row_sqnorms(ix->dX, ix->ld, n, d, ix->dXnorm2.p, c.dBNorm.p, c.stream));
    VT_HIP(hipMemcpyAsync(&bits, c.dBNorm.p, sizeof(bits), hipMemcpyDeviceToHost, c.stream));
    VT_HIP(hipStreamSynchronize(c.stream));
    std::memcpy(&ix->max_sqnorm, &bits, sizeof(double));
  }

  return VT_OK;
}

}  // namespace
