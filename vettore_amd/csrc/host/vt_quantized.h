// vt_quantized.h -- quantized_search on one shard (collection.ex:276-295): K4 / K4h + rerank as one device chain, groups of eight per sweep of the sign bits
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// What one shard of a multi-shard handle contributes to a staged search in ONE round: its
// own candidates under each generator's cutting keys (ascending) and the exact-rerank entries
// of all of them -- uncut, because which of them belong to the handle-wide candidate set is
// only known once the shards' lists meet (staged_merge).
struct LocalStages {
  std::vector<std::vector<vt::Entry>> gens;
  std::vector<vt::Entry> final_;
};

void entries_of_block(const ResultBlock *b, std::vector<vt::Entry> &out) { out.assign(b->e, b->e + b->count); }

// collection.ex:276-295 on a shard whose ranks (strict) and sign bits are current.
// `local` (multi-shard handles; candidates <= kMaxFusedK): nothing is cut to `limit` and no
// hit list is built -- the shard's candidate and rerank entries go to *local.
// K4h for the query whose sign bits are in c.dQbits: distance column + histogram, then the rows up
// to the k1-th distance as an unsorted key list in c.dPartKeys / c.dPartPay (c.dHamCount entries,
// at most kHammingListCap; more ties than that raise kStatusRetry in c.dStatus).  Queued, not waited for.
constexpr uint32_t kHammingListCap = 65536, kHammingHistStride = 8192;
int hamming_hist_collect(Shard *ix, Ctx &c, uint32_t k1, bool *timed) {
  const uint32_t d = (uint32_t)ix->dim, words = (d + 63) / 64;
  VT_TRY(c.dDist16.ensure(((size_t)std::max<uint32_t>(ix->cap, ix->n) + 7) / 8 * 8));
  VT_TRY(c.dHamHist.ensure(2 * kHammingHistStride));
  VT_TRY(c.dHamCount.ensure(1));
  VT_TRY(c.dPartKeys.ensure(kHammingListCap));
  VT_TRY(c.dPartPay.ensure(kHammingListCap));
  if (!c.ham_ready || c.ham_dirty) {
    VT_HIP(hipMemsetAsync(c.dHamHist.p, 0, 2 * kHammingHistStride * sizeof(uint32_t), c.stream));
    c.ham_ready = true;
  }
  c.ham_dirty = true;  // until this query's collect pass has been queued
  vt::HammingHistArgs h{};
  h.bits = ix->dBits.p;
  h.qbits = c.dQbits;
  h.n = ix->n;
  h.words = words;
  h.pairs = (words + 1) / 2;
  h.d = d;
  h.dist = c.dDist16.p;
  h.hist = c.dHamHist.p + c.ham_parity * kHammingHistStride;
  h.list_count = c.dHamCount.p;
  const uint32_t blocks = c.grid_for((ix->n + 63) / 64, vt::hamming_hist_lds_bytes(d), c.hamming_blocks_per_cu);
  if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
  VT_HIP(vt::launch_hamming_dist(h, blocks, c.stream));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
  *timed = c.profiling;
  vt::HammingCollectArgs g{};
  g.dist = c.dDist16.p;
  g.id_rank = ix->dRank.p;
  g.n = ix->n;
  g.d = d;
  g.k = k1;
  g.hist = h.hist;
  g.hist_next = c.dHamHist.p + (c.ham_parity ^ 1u) * kHammingHistStride;
  g.list_count = c.dHamCount.p;
  g.keys = c.dPartKeys.p;
  g.pay = c.dPartPay.p;
  g.cap = kHammingListCap;
  g.status = c.dStatus.p;
  VT_HIP(vt::launch_hamming_collect(g, (uint32_t)c.num_cus * 4, c.stream));
  c.ham_parity ^= 1u;
  c.ham_dirty = false;
  return VT_OK;
}

// binary_top_k (search.rs:76-92) on the device for the query in c.dQbits: the k1 <= kMaxFusedK
// nearest rows, sorted, into the device block `dst` (whose Entry.row column the next stage
// gathers by).  use_hist: K4h (stream + histogram + threshold collect), else K4 (fused top-k).
// No select here takes the status word: a raised flag stays in c.dStatus for the call's last select.
int hamming_stage_dev(Shard *ix, Ctx &c, uint32_t k1, bool use_hist, ResultBlock *dst, bool *timed) {
  const uint32_t d = (uint32_t)ix->dim, words = (d + 63) / 64;
  if (use_hist) {
    VT_TRY(hamming_hist_collect(ix, c, k1, timed));
    VT_HIP(vt::launch_select(c.dPartKeys.p, c.dPartPay.p, kHammingListCap, k1, 0, 0, nullptr, dst, c.dSelKeys.p, c.dSelPay.p,
                             c.stream, c.dHamCount.p));
    return VT_OK;
  }
  const uint32_t blocks = c.grid_for((ix->n + 63) / 64, vt::hamming_lds_bytes(k1), c.hamming_blocks_per_cu);
  const uint32_t waves = vt::scan_lists(blocks);
  VT_TRY(c.dPartKeys.ensure((size_t)waves * k1));
  VT_TRY(c.dPartPay.ensure((size_t)waves * k1));
  vt::HammingArgs h{};
  h.bits = ix->dBits.p;
  h.qbits = c.dQbits;
  h.id_rank = ix->dRank.p;
  h.n = ix->n;
  h.words = words;
  h.pairs = (words + 1) / 2;
  h.d = d;
  h.k = k1;
  h.part_keys = c.dPartKeys.p;
  h.part_pay = c.dPartPay.p;
  if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
  VT_HIP(vt::launch_hamming(h, blocks, c.stream));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
  *timed = c.profiling;
  VT_HIP(vt::launch_select(c.dPartKeys.p, c.dPartPay.p, waves * k1, k1, 0, 0, nullptr, dst, c.dSelKeys.p, c.dSelPay.p, c.stream));
  return VT_OK;
}

int quantized_ready(Shard *ix, Ctx &c, const float *query, size_t n, size_t candidates, size_t limit, vt_hits **out,
                    LocalStages *local = nullptr) {
  // collection.ex:276-295: prepare_query validates the query against the
  // collection's dimension; an empty store yields no candidates.
  VT_TRY(validate_vector(query, n, ix->dim));
  if (ix->n == 0 || candidates == 0 || limit == 0) return empty_hits(out);
  const uint32_t d = (uint32_t)ix->dim;
  const uint32_t words = (d + 63) / 64;
  uint32_t qnz = 0;
  // (the distance pass fetches the query's 12 words per block, the rerank its floats for a few dozen blocks: no copy)
  VT_TRY(upload_query(c, query, n, &qnz, true));
  const size_t ncand = std::min<size_t>(candidates, ix->n);
  const size_t keep = local ? ncand : limit;
  // K4h needs integer bins in LDS, one fused select, and enough rows to be worth two passes
  const bool hist_ok = ncand <= (size_t)vt::kSelListMax && d <= vt::kHammingHistMaxDim &&
                       (ix->n >= 16384 || ncand > (size_t)vt::kMaxFusedK);
  auto run = [&](bool use_hist) -> int {
  std::vector<vt::Entry> entries, first;
  bool first_in_block = false;
  const uint32_t *gather = nullptr;
  uint32_t gather_stride = 1;
  bool timed_hamming = false;
  auto copy_first_block = [&]() -> int {  // queued behind the select that fills c.dStage[0]
    if (!local) return VT_OK;
    VT_TRY(c.hFirst.ensure(1));
    VT_HIP(hipMemcpyAsync(c.hFirst.p, c.dStage.p, sizeof(ResultBlock), hipMemcpyDeviceToHost, c.stream));
    first_in_block = true;
    return VT_OK;
  };
  if (ncand <= (size_t)vt::kMaxFusedK) {
    // stage 1 stays on the device (K4h as a pure stream, or K4's fused top-k): its winners land in
    // a device block whose Entry.row column is the gather list of stage 2 (no host round trip)
    VT_TRY(c.dStage.ensure(1));
    VT_TRY(hamming_stage_dev(ix, c, (uint32_t)ncand, use_hist, c.dStage.p, &timed_hamming));
    VT_TRY(copy_first_block());
    gather = &c.dStage.p->e[0].row;
    gather_stride = sizeof(vt::Entry) / sizeof(uint32_t);
  } else if (use_hist) {
    if (local) return VT_ERR_ARGUMENT;  // callers keep one-round searches to candidates <= kMaxFusedK
    // up to 4 096 candidates (limit * 10 for limit <= 409): the exact candidate SET as a
    // device list -- stage 2 orders by its own keys, so this one need not be sorted
    const uint32_t k1 = (uint32_t)ncand;
    VT_TRY(hamming_hist_collect(ix, c, k1, &timed_hamming));
    VT_TRY(c.dListKeys.ensure(k1));
    VT_TRY(c.dListPay.ensure(k1));
    VT_HIP(vt::launch_select_list(c.dPartKeys.p, c.dPartPay.p, kHammingListCap, c.dHamCount.p, k1, c.dListKeys.p, c.dListPay.p,
                                  c.stream));
    gather = &c.dListPay.p->row;
    gather_stride = sizeof(vt::Payload) / sizeof(uint32_t);
  } else {
    // stage 1: binary_top_k (search.rs:76-92), candidate rows via the host
    std::vector<vt::Entry> cand;
    VT_TRY(run_hamming(c, ix->dBits.p, c.dQbits, ix->dRank.p, ix->n, d, candidates, cand, true));
    if (local) first = cand;
    std::vector<uint32_t> rows(cand.size());
    for (size_t i = 0; i < cand.size(); ++i) rows[i] = cand[i].row;
    VT_TRY(c.dRows.ensure(rows.size()));
    VT_HIP(hipMemcpyAsync(c.dRows.p, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
    VT_HIP(hipStreamSynchronize(c.stream));  // `rows` is pageable and dies with this scope
    gather = c.dRows.p;
  }
  // stage 2: vector_top_k over the candidates (search.rs:38-73)
  if (ix->metric == VT_COSINE) {
    VT_TRY(c.dCandKeys.ensure(ncand));
    VT_TRY(c.dCandPay.ensure(ncand));
    vt::CosineRerankArgs a{};
    a.X = ix->dX;
    a.stride = ix->ld;
    a.q = c.qsrc;
    a.id_rank = ix->dRank.p;
    a.gather = gather;
    a.gather_stride = gather_stride;
    a.n = (uint32_t)ncand;
    a.d = d;
    a.out_keys = c.dCandKeys.p;
    a.out_pay = c.dCandPay.p;
    a.status = c.dStatus.p;
    VT_HIP(vt::launch_cosine_rerank(a, c.stream));
    VT_TRY(collect_from_keys(c, c.dCandKeys.p, c.dCandPay.p, (uint32_t)ncand, keep, entries));
  } else {
    ScanJob j{};
    j.X = ix->dX;
    j.stride = ix->ld;
    j.id_rank = ix->dRank.p;
    j.gather = gather;
    j.gather_stride = gather_stride;
    j.n = (uint32_t)ncand;
    j.d = d;
    j.metric = ix->metric;
    j.order = ix->order;
    j.q_nonzero = qnz;
    VT_TRY(run_scan(c, j, keep, entries, false));
  }
  if (timed_hamming) {
    float ms = 0.f;
    VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    c.prof.hamming_launches += 1;
    c.prof.hamming_ms += ms;
    c.prof.hamming_bytes += (uint64_t)ix->n * words * 8;
  }
  if (local) {
    if (first_in_block) entries_of_block(c.hFirst.p, first);  // (every path above ends in a stream sync)
    local->gens.assign(1, std::move(first));
    local->final_ = std::move(entries);
    return VT_OK;
  }
  return make_hits(ix, entries, out);
  };
  int rc = run(hist_ok);
  if (rc == kRetryInternal) rc = run(false);  // more ties at the k-th distance than the list holds
  return rc;
}


// ---- several quantized searches in ONE sweep of the bit matrix ------------------------------
// collection.ex:276-295 for up to kHammingMultiMax queries at once, the way K1m carries several
// plain searches: the 96 bytes of sign bits per row (0.96 GB at N = 10 M) are read once, every
// stage behind them runs once with the queries on grid.y -- distance columns + histograms,
// threshold collect, select, exact rerank (f64 cosine, or K1 on the gathered rows), select --
// and one wait ends the group.  Each query's hits are what its own quantized_search returns,
// bit for bit (same kernels, same keys).  kRetryInternal: something only the single-query path
// sorts out (more ties at a k-th distance than a list holds, a metric overflow in the rerank).
bool quantized_group_applies(const Shard *ix, size_t candidates, size_t limit) {
  const uint32_t d = (uint32_t)ix->dim;
  const size_t ncand = std::min<size_t>(candidates, ix->n);
  // (jaccard: the rerank's non-zero count of the query is one launch argument, so those go query by query -- said
  // here, before a sweep of the bit matrix has been spent on finding out)
  return ix->metric != VT_JACCARD && ix->n >= 16384 && ncand >= 1 && ncand <= (size_t)vt::kMaxFusedK && limit >= 1 && d <= vt::kHammingHistMaxDim &&
         vt::hamming_multi_lds_bytes(d, (d + 63) / 64, 2) <= 64 * 1024 &&
         (ix->metric == VT_COSINE ? (size_t)2 * ((d + 3) / 4 * 4) * 4 <= 160 * 1024 : vt::scan_lds_bytes(d, (uint32_t)std::min<size_t>(limit, ncand)) != 0);
}
// queries per sweep: what the nq histograms leave room for in 64 KiB of LDS
uint32_t quantized_group_size(const Shard *ix) {
  const uint32_t d = (uint32_t)ix->dim;
  uint32_t nq = vt::kHammingMultiMax;
  while (nq > 1 && vt::hamming_multi_lds_bytes(d, (d + 63) / 64, nq) > 64 * 1024) nq -= 1;
  return nq;
}

// Where a group's upload and its results live: several groups of one call are queued behind each
// other -- the device scratch is reused in stream order, only what the HOST writes or reads has a
// region per group -- and waited for once (quantized_batch_ready).
struct QuantizedGroupSlot {
  uint32_t slot, nslots;
  size_t up_floats, res_bytes;  // per-slot sizes (the same for every group of a call)
};
QuantizedGroupSlot quantized_group_slot(const Shard *ix, uint32_t slot, uint32_t nslots) {
  const uint32_t words = ((uint32_t)ix->dim + 63) / 64, pairs = (words + 1) / 2;
  const size_t up = (size_t)vt::kHammingMultiMax * ix->ld + 2 * (size_t)vt::kHammingMultiMax * 2 * pairs + vt::kHammingMultiMax;
  return QuantizedGroupSlot{slot, nslots, (up + 63) / 64 * 64, (size_t)vt::kHammingMultiMax * vt::kMaxFusedK * sizeof(vt::Entry) + 64};
}

int quantized_group_finish(Shard *ix, Ctx &c, const QuantizedGroupSlot &gs, const std::vector<size_t> &which, uint32_t k2,
                           vt_hits **out) {
  const uint32_t nq = (uint32_t)which.size();
  const size_t ent_bytes = (size_t)nq * k2 * sizeof(vt::Entry);
  const unsigned char *res = c.hBig.p + (size_t)gs.slot * gs.res_bytes;
  const vt::Entry *hOut = reinterpret_cast<const vt::Entry *>(res);
  const uint32_t *hOutCount = reinterpret_cast<const uint32_t *>(res + ent_bytes);
  const int status = *reinterpret_cast<const int *>(res + ent_bytes + 32);
  if (status != 0) return kRetryInternal;  // a tie list overflowed / a rerank overflowed: one by one, each reports its own
  for (uint32_t i = 0; i < nq; ++i) {
    const uint32_t got = std::min<uint32_t>(hOutCount[i], k2);
    std::vector<vt::Entry> entries(hOut + (size_t)i * k2, hOut + (size_t)i * k2 + got);
    VT_TRY(make_hits(ix, entries, &out[which[i]]));
  }
  return VT_OK;
}

// `defer`: everything is queued, nothing waited for -- the caller waits for the stream and then
// calls quantized_group_finish for the slot.
int quantized_group(Shard *ix, Ctx &c, const float *queries, const std::vector<size_t> &which, size_t candidates, size_t limit,
                    vt_hits **out, uint32_t slot = 0, uint32_t nslots = 1, bool defer = false) {
  if (which.size() < 2) return kRetryInternal;  // (groups are of two or more: grid.y is what tells the selects apart)
  const QuantizedGroupSlot gs = quantized_group_slot(ix, slot, nslots);
  const uint32_t d = (uint32_t)ix->dim, ld = ix->ld, n = ix->n;
  const uint32_t words = (d + 63) / 64, pairs = (words + 1) / 2;
  const uint32_t nq = (uint32_t)which.size();
  const uint32_t k1 = (uint32_t)std::min<size_t>(candidates, n);
  const uint32_t k2 = (uint32_t)std::min<size_t>(limit, k1);
  constexpr uint32_t kListCap = 8192;
  const uint32_t hist_stride = (d + 1 + 63) / 64 * 64;
  const uint32_t dist_stride = (std::max<uint32_t>(ix->cap, n) + 7) / 8 * 8;
  // ONE upload: the queries (f32, padded rows), their sign bits (all eight slots, the unused ones
  // zero) and the candidate counts K1's batch mode wants sit behind each other in one pinned
  // block and one device block (each async copy costs ~10 us of a 0.4-ms group)
  const size_t q_floats = (size_t)nq * ld;
  const size_t bit_words = (size_t)vt::kHammingMultiMax * 2 * pairs;  // u64, 8-byte aligned behind ld-multiples of floats
  const size_t up_floats = q_floats + 2 * bit_words + vt::kHammingMultiMax;
  VT_TRY(c.dBQ.ensure((size_t)nslots * gs.up_floats));
  VT_TRY(c.hBQ.ensure((size_t)nslots * gs.up_floats));
  float *const hq = c.hBQ.p + (size_t)slot * gs.up_floats, *const dq = c.dBQ.p + (size_t)slot * gs.up_floats;
  VT_TRY(c.dDist16.ensure((size_t)vt::kHammingMultiMax * dist_stride));  // dist[row][8]
  VT_TRY(c.dHamHist.ensure(std::max<size_t>((size_t)nq * hist_stride, 2 * 8192)));
  VT_TRY(c.dHamCount.ensure(vt::kHammingMultiMax));
  VT_TRY(c.dPartKeys.ensure((size_t)nq * kListCap));
  VT_TRY(c.dPartPay.ensure((size_t)nq * kListCap));
  VT_TRY(c.dStageB.ensure(nq));
  // results through the host mapping (no D2H copies): [nq][k2] entries, then nq counts, then the status word
  const size_t res_bytes = (size_t)nslots * gs.res_bytes;
  if (!c.dBigMapped || c.hBig.count < res_bytes) {
    VT_TRY(c.hBig.ensure(std::max<size_t>(res_bytes, 16 + (size_t)vt::kSelListMax * sizeof(vt::Entry))));
    VT_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&c.dBigMapped), c.hBig.p, 0));
  }
  const size_t ent_bytes = (size_t)nq * k2 * sizeof(vt::Entry);
  unsigned char *const dres = c.dBigMapped + (size_t)slot * gs.res_bytes;
  vt::Entry *dOut = reinterpret_cast<vt::Entry *>(dres);
  uint32_t *dOutCount = reinterpret_cast<uint32_t *>(dres + ent_bytes);
  int *hStatus = reinterpret_cast<int *>(c.hBig.p + (size_t)slot * gs.res_bytes + ent_bytes + 32);
  std::vector<uint32_t> qnz(nq, 0);
  std::memset(hq, 0, up_floats * sizeof(float));
  uint64_t *hbits = reinterpret_cast<uint64_t *>(hq + q_floats);
  uint32_t *hcounts = reinterpret_cast<uint32_t *>(hq + q_floats + 2 * bit_words);
  for (uint32_t i = 0; i < nq; ++i) {
    const float *q = queries + which[i] * d;
    std::memcpy(hq + (size_t)i * ld, q, (size_t)d * sizeof(float));
    uint64_t *w = hbits + (size_t)i * 2 * pairs;
    for (uint32_t j = 0; j < d; ++j) {
      qnz[i] += q[j] != 0.0f ? 1u : 0u;
      if (q[j] >= 0.0f) w[j / 64] |= 1ull << (j % 64);  // distances.rs:413-423
    }
    hcounts[i] = k1;
  }
  VT_HIP(hipMemcpyAsync(dq, hq, up_floats * sizeof(float), hipMemcpyHostToDevice, c.stream));
  const uint64_t *dbits = reinterpret_cast<const uint64_t *>(dq + q_floats);
  const uint32_t *dcounts = reinterpret_cast<const uint32_t *>(dq + q_floats + 2 * bit_words);
  VT_HIP(hipMemsetAsync(c.dHamHist.p, 0, (size_t)nq * hist_stride * sizeof(uint32_t), c.stream));
  c.ham_dirty = true;  // (the single-query path's two alternating histograms live in the same buffer)
  vt::HammingMultiArgs h{};
  h.bits = ix->dBits.p;
  h.qbits = dbits;
  h.n = n;
  h.words = words;
  h.pairs = pairs;
  h.d = d;
  h.nq = nq;
  h.dist = c.dDist16.p;
  h.dist_stride = dist_stride;
  h.hist = c.dHamHist.p;
  h.hist_stride = hist_stride;
  h.list_count = c.dHamCount.p;
  // (more waves per CU than the single pass keeps: eight queries' scalar loads and popcounts per tile
  // want their latency hidden -- 0.230 ms at 2 blocks per CU, 0.210 at 4, N = 10 M)
  const uint32_t blocks = c.grid_for((n + 63) / 64, vt::hamming_multi_lds_bytes(d, words, nq), std::max(4, c.hamming_blocks_per_cu));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
  VT_HIP(vt::launch_hamming_dist_multi(h, blocks, c.stream));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
  vt::HammingCollectArgs g{};
  g.dist = c.dDist16.p;
  g.id_rank = ix->dRank.p;
  g.n = n;
  g.d = d;
  g.k = k1;
  g.hist = c.dHamHist.p;
  g.hist_next = nullptr;
  g.list_count = c.dHamCount.p;
  g.keys = c.dPartKeys.p;
  g.pay = c.dPartPay.p;
  g.cap = kListCap;
  g.status = c.dStatus.p;
  g.dist_stride = dist_stride;
  g.hist_stride = hist_stride;
  VT_HIP(vt::launch_hamming_collect_multi(g, (uint32_t)c.num_cus * 4, nq, c.stream));
  VT_HIP(vt::launch_select_lists(c.dPartKeys.p, c.dPartPay.p, nq, kListCap, c.dHamCount.p, k1, c.dStageB.p,
                                 (uint32_t)sizeof(ResultBlock), c.stream));
  // stage 2: vector_top_k over each query's candidates (search.rs:38-73)
  const uint32_t gather_qstride = (uint32_t)(sizeof(ResultBlock) / sizeof(uint32_t));
  if (ix->metric == VT_COSINE) {
    VT_TRY(c.dCandKeys.ensure((size_t)nq * k1));
    VT_TRY(c.dCandPay.ensure((size_t)nq * k1));
    vt::CosineRerankArgs a{};
    a.X = ix->dX;
    a.stride = ix->ld;
    a.q = dq;
    a.id_rank = ix->dRank.p;
    a.gather = &c.dStageB.p->e[0].row;
    a.gather_stride = sizeof(vt::Entry) / sizeof(uint32_t);
    a.n = k1;
    a.d = d;
    a.out_keys = c.dCandKeys.p;
    a.out_pay = c.dCandPay.p;
    a.status = c.dStatus.p;
    a.q_stride = ld;
    a.gather_qstride = gather_qstride;
    VT_HIP(vt::launch_cosine_rerank_batch(a, nq, c.stream));
    VT_HIP(vt::launch_batch_select(c.dCandKeys.p, c.dCandPay.p, nq, k1, k2, dOut, dOutCount, c.stream));
  } else {
    constexpr uint32_t kBlocksPerQuery = 2;
    VT_TRY(c.dCandKeys.ensure((size_t)nq * kBlocksPerQuery * k2));
    VT_TRY(c.dCandPay.ensure((size_t)nq * kBlocksPerQuery * k2));
    // the stage-1 blocks are 4 112 bytes apart = 257 entries of 16: K1's batch mode walks query y's
    // list at gather + y * batch_cap * gather_stride
    static_assert(sizeof(ResultBlock) == 257 * sizeof(vt::Entry), "stage blocks as K1 batch lists");
    vt::ScanArgs sa{};
    sa.X = ix->dX;
    sa.stride = ix->ld;
    sa.q = dq;
    sa.id_rank = ix->dRank.p;
    sa.gather = &c.dStageB.p->e[0].row;
    sa.gather_stride = sizeof(vt::Entry) / sizeof(uint32_t);
    sa.n = 257;
    sa.d = d;
    sa.metric = ix->metric;
    sa.order = ix->order;
    sa.k = k2;
    sa.part_keys = c.dCandKeys.p;
    sa.part_pay = c.dCandPay.p;
    sa.status = c.dStatus.p;
    sa.batch_counts = dcounts;
    sa.batch_cap = 257;
    // (jaccard needs the query's non-zero count: one value per launch, so those go query by query)
    if (ix->metric == VT_JACCARD) return kRetryInternal;
    VT_HIP(vt::launch_scan_batch(sa, kBlocksPerQuery, nq, c.stream));
    VT_HIP(vt::launch_batch_select(c.dCandKeys.p, c.dCandPay.p, nq, kBlocksPerQuery * k2, k2, dOut, dOutCount, c.stream));
  }
  VT_HIP(hipMemcpyAsync(hStatus, c.dStatus.p, sizeof(int), hipMemcpyDeviceToHost, c.stream));  // (pinned: stays asynchronous)
  VT_HIP(hipMemsetAsync(c.dStatus.p, 0, sizeof(int), c.stream));
  if (defer) return VT_OK;
  VT_HIP(hipStreamSynchronize(c.stream));
  if (c.profiling) {
    float ms = 0.f;
    VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    c.prof.hamming_launches += 1;
    c.prof.hamming_ms += ms;
    c.prof.hamming_bytes += (uint64_t)n * words * 8;
    c.prof.hamming_queries += nq;
  }
  return quantized_group_finish(ix, c, gs, which, k2, out);
}

// quantized_search for nq queries (rows of `queries`): groups of up to eight share a sweep; what
// the groups cannot take (one query left over, a shape outside the group path, a retry) goes
// through quantized_ready one by one.  Ranks strictly current, sign bits current.
int quantized_batch_ready(Shard *ix, Ctx &c, const float *queries, size_t nq, size_t d, size_t candidates, size_t limit,
                          vt_hits **out) {
  for (size_t i = 0; i < nq; ++i) VT_TRY(validate_vector(queries + i * d, d, ix->dim));
  if (ix->n == 0 || candidates == 0 || limit == 0) {
    for (size_t i = 0; i < nq; ++i) VT_TRY(empty_hits(&out[i]));
    return VT_OK;
  }
  std::vector<char> done(nq, 0);
  if (nq >= 2 && quantized_group_applies(ix, candidates, limit)) {
    const uint32_t per = quantized_group_size(ix);
    std::vector<std::vector<size_t>> groups;
    for (size_t g0 = 0; g0 < nq && per >= 2; g0 += per) {
      std::vector<size_t> which;
      for (size_t i = g0; i < std::min(nq, g0 + per); ++i) which.push_back(i);
      if (which.size() < 2) break;
      groups.push_back(std::move(which));
    }
    auto settle = [&](const std::vector<size_t> &which, int st) -> int {
      if (st == VT_OK) {
        for (size_t i : which) done[i] = 1;
      } else if (st == kRetryInternal) {
        for (size_t i : which) {
          delete out[i];
          out[i] = nullptr;
        }
      } else {
        return st;
      }
      return VT_OK;
    };
    // Several groups: all of them are queued behind each other (the device scratch is reused in
    // stream order; uploads and results have a region per group) and waited for ONCE -- a group's
    // host side (staging 8 queries and their sign bits, the wait, 8 hit lists: 0.1 of its 0.38 ms)
    // then runs while the device is busy with the groups around it.  (Not while profiling: the
    // stage timing keeps one pair of events per context.)
    if (groups.size() >= 2 && groups.size() <= 32 && !c.profiling && !vt::env::on(vt::env::NO_GROUP_PIPELINE)) {
      // (r05) ... on TWO contexts when a second one is to be had: even groups here, odd groups there, each stream its
      // own slots.  A group is a sweep of the bits (0.20 ms for eight queries) and a tail of small kernels -- collect,
      // list select, rerank of 8 x `candidates` rows, select: 0.07 ms -- that then runs beside the other stream's sweep
      // instead of in front of it (the one-stream form of r04 is what runs when no second context is to be had).
      const uint32_t ng = (uint32_t)groups.size();
      SpareCtxLease spare(ix);
      if (spare.c) VT_TRY(reader_ready(ix, *spare.c));
      Ctx *cx[2] = {&c, spare.c ? spare.c : &c};
      const uint32_t lanes = spare.c ? 2u : 1u;
      auto ctx_of = [&](uint32_t g) -> Ctx & { return *cx[g % lanes]; };
      auto slot_of = [&](uint32_t g) { return g / lanes; };
      auto slots_of = [&](uint32_t g) { return (ng - (g % lanes) + lanes - 1) / lanes; };  // groups on this group's context
      auto drain = [&]() {
        (void)hipStreamSynchronize(cx[0]->stream);
        if (lanes == 2) (void)hipStreamSynchronize(cx[1]->stream);
      };
      const uint32_t k2 = (uint32_t)std::min<size_t>(limit, std::min<size_t>(candidates, ix->n));
      std::vector<int> queued(groups.size(), VT_OK);
      for (uint32_t g = 0; g < ng; ++g) {
        queued[g] = quantized_group(ix, ctx_of(g), queries, groups[g], candidates, limit, out, slot_of(g), slots_of(g), true);
        if (queued[g] != VT_OK && queued[g] != kRetryInternal) {
          const std::string why = g_last_error;
          drain();
          (void)hipGetLastError();
          g_last_error = why;
          return queued[g];
        }
      }
      VT_HIP(hipStreamSynchronize(cx[0]->stream));
      if (lanes == 2) VT_HIP(hipStreamSynchronize(cx[1]->stream));
      for (uint32_t g = 0; g < ng; ++g) {
        const int st = queued[g] == VT_OK ? quantized_group_finish(ix, ctx_of(g), quantized_group_slot(ix, slot_of(g), slots_of(g)),
                                                                   groups[g], k2, out)
                                          : queued[g];
        VT_TRY(settle(groups[g], st));
      }
    } else {
      for (const auto &which : groups) {
        VT_TRY(settle(which, quantized_group(ix, c, queries, which, candidates, limit, out)));
      }
    }
  }
  for (size_t i = 0; i < nq; ++i)
    if (!done[i]) VT_TRY(quantized_ready(ix, c, queries + i * d, d, candidates, limit, &out[i]));
  return VT_OK;
}

}  // namespace
