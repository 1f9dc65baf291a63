// vt_funnel.h -- funnel_search on one shard (collection.ex:245-260): the chained stages, groups of eight per sweep of the prefixes (K6bm / K1p / the bit column)
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// collection.ex:245-260 on a shard whose ranks are strictly current.
int funnel_ready(Shard *ix, Ctx &c, const float *query, size_t n, const size_t *stages, size_t nstages,
                 size_t candidates, size_t limit, vt_hits **out, LocalStages *local = nullptr) {
  // collection.ex:245-260: prepare_query validates the query against the
  // collection; stages are prefix lengths 1..dimensions (collection.ex:905-913)
  VT_TRY(validate_vector(query, n, ix->dim));
  if (nstages == 0) return VT_ERR_PREFIX;
  for (size_t i = 0; i < nstages; ++i)
    if (stages[i] == 0 || stages[i] > n) return VT_ERR_PREFIX;
  if (ix->n == 0 || candidates == 0 || limit == 0) return empty_hits(out);
  uint32_t qnz_full = 0;
  // (float hamming / jaccard: the query's non-zero bits ride along -- the stage over all rows reads the bit column)
  // (stage 1 fetches a prefix per block, the later stages and the rerank a few dozen blocks' worth: no copy)
  VT_TRY(upload_query(c, query, n, &qnz_full, pattern_metric(ix->metric) ? 2 : 0));
  std::vector<vt::Entry> entries;
  // (`local`: the rerank keeps every candidate -- see LocalStages)
  if (funnel_fits_device(ix, stages, nstages, candidates, local ? candidates : limit)) {
    // the whole funnel as one chain of kernels: each stage's winners stay in a
    // device block whose row column is the next stage's gather list; one wait
    VT_TRY(c.dStage.ensure(2));
    const ResultBlock *src = nullptr;
    uint32_t count = ix->n;
    for (size_t i = 0; i < nstages; ++i) {
      uint32_t nz = 0;
      for (size_t j = 0; j < stages[i]; ++j) nz += query[j] != 0.0f ? 1u : 0u;
      const uint32_t want = (uint32_t)std::min<size_t>(candidates, count);
      ResultBlock *dst = c.dStage.p + (i & 1);
      VT_TRY(funnel_stage_dev(ix, c, query, (uint32_t)stages[i], src, count, want, nz, dst, false));
      if (local && i == 0) {
        VT_TRY(c.hFirst.ensure(1));
        VT_HIP(hipMemcpyAsync(c.hFirst.p, dst, sizeof(ResultBlock), hipMemcpyDeviceToHost, c.stream));
      }
      src = dst;
      count = want;
    }
    // exact_rerank on the full vectors (collection.ex:821-851)
    const uint32_t want = local ? count : (uint32_t)std::min<size_t>(limit, count);
    VT_TRY(funnel_stage_dev(ix, c, query, (uint32_t)ix->dim, src, count, want, qnz_full, c.dResMapped, true));
    VT_HIP(hipStreamSynchronize(c.stream));
    VT_TRY(c.settle_prefix_profile());
    if (c.hRes.p->status == VT_ERR_OVERFLOW) return VT_ERR_OVERFLOW;
    entries.assign(c.hRes.p->e, c.hRes.p->e + c.hRes.p->count);
    if (local) {
      local->gens.resize(1);
      entries_of_block(c.hFirst.p, local->gens[0]);
      local->final_ = std::move(entries);
      return VT_OK;
    }
    return make_hits(ix, entries, out);
  }
  std::vector<uint32_t> rows;
  std::vector<vt::Entry> first;
  VT_TRY(funnel_rows(ix, c, query, stages, nstages, candidates, rows, local ? &first : nullptr));
  if (local) {
    if (!rows.empty()) VT_TRY(funnel_stage(ix, c, query, (uint32_t)ix->dim, rows, false, rows.size(), qnz_full, entries));
    local->gens.assign(1, std::move(first));
    local->final_ = std::move(entries);
    return VT_OK;
  }
  if (rows.empty()) return empty_hits(out);
  // exact_rerank on the full vectors (collection.ex:821-851)
  VT_TRY(funnel_stage(ix, c, query, (uint32_t)ix->dim, rows, false, limit, qnz_full, entries));
  return make_hits(ix, entries, out);
}


// ---- several funnel searches per sweep of the prefixes (cosine collections) ------------------
// collection.ex:245-260 for up to kCosineMultiMax queries at once, the way quantized searches share
// a sweep of the sign bits: stage 1 -- the f64 cosine over the first stages[0] coordinates of
// EVERY row, 0.8 ms of a 0.88-ms funnel search at N = 10 M -- reads the prefixes once for the whole
// group (cosine_scan_multi_kernel).  The scores are exact, so a threshold needs no margin: tau_q =
// the rank-th best score of a sample of the rows (~6 x candidates rows pass), every (query, row)
// reaching it is listed, the lists are cut to their `candidates` best by the batched select, and
// a query whose list came out short (or overflowed) takes the single path.  Later stages and the
// exact rerank run once with the queries on grid.y (cosine_rerank_kernel); one wait.  Each
// query's hits are what its own funnel_search returns, bit for bit.
//
// Collections of the dot / L2 / L1 / Linf families go the same way with their own arithmetic: stage 1 is K1p
// (prefix_multi_kernel: K1's chunked f32 sums over the prefix for up to eight queries per sweep), the threshold is
// taken on -rank_value (exact negation: the list is cut by the very order the keys sort in, so a list of at least
// `candidates` rows IS the single path's stage), later stages and the rerank are K1's batch mode over the
// candidates (queries on grid.y), the arithmetic the single path's scan_stage_dev runs.
bool funnel_group_applies(const Shard *ix, const size_t *stages, size_t nstages, size_t candidates, size_t limit) {
  if (nstages == 0) return false;
  if (ix->metric != VT_COSINE && !vt::prefix_multi_supports(ix->metric)) return false;
  const size_t k1 = std::min<size_t>(candidates, ix->n);
  if (ix->n < 16384 || k1 == 0 || k1 > (size_t)vt::kMaxFusedK || limit == 0) return false;
  if (ix->metric != VT_COSINE) return vt::scan_lds_bytes((uint32_t)ix->dim, (uint32_t)k1) != 0;
  return (size_t)2 * (((size_t)ix->dim + 3) / 4 * 4) * 4 <= 160 * 1024;  // the rerank keeps row + query in LDS
}

// A group in flight: funnel_group_queue leaves everything queued on the context's stream, funnel_group_finish waits for
// it and files the lists.  Between the two the context's pinned blocks belong to the group (one group per context at a
// time); the views below point into them.
struct FunnelGroupRun {
  std::vector<size_t> which;
  uint32_t nq = 0, k1 = 0, k2 = 0, n = 0, d1 = 0;
  bool cosine = false, as_scan = false;
  const vt::Entry *hOut = nullptr;
  const uint32_t *hOutCount = nullptr;
  const int *hStatus = nullptr;
  const uint32_t *hListCount = nullptr;
  const float *hTau = nullptr;
  const uint64_t *hLastKey = nullptr;
};

int funnel_group_queue(Shard *ix, Ctx &c, FunnelGroupRun &run, const float *queries, const std::vector<size_t> &which,
                       const size_t *stages, size_t nstages, size_t candidates, size_t limit, bool as_scan) {
  const uint32_t d = (uint32_t)ix->dim, ld = ix->ld, n = ix->n;
  const uint32_t nq = (uint32_t)which.size();
  const uint32_t d1 = (uint32_t)stages[0], ldq = vt::padded_dim(d1);
  const uint32_t k1 = (uint32_t)std::min<size_t>(candidates, n);
  const uint32_t k2 = (uint32_t)std::min<size_t>(limit, k1);
  constexpr uint32_t kListCap = 8192, kSampleTiles = 1024;
  const uint32_t ntiles = (n + 63) / 64;
  const uint32_t sstride = (ntiles + kSampleTiles - 1) / kSampleTiles;
  const uint32_t stiles = (ntiles + sstride - 1) / sstride;
  const uint32_t sample_rows = stiles * 64;
  // about six lists' worth of rows pass (the count is Gamma(rank)-distributed around its mean:
  // at rank >= 6 a list shorter than `candidates` is a 1e-4 event; it costs a single search)
  uint32_t rank = (uint32_t)std::ceil(6.0 * k1 * std::min(1.0, (double)sample_rows / (double)n));
  rank = std::max<uint32_t>(6, std::min<uint32_t>(rank, std::min<uint32_t>(sample_rows, n)));
  // (r05) the sample pass files only the best score of each of its 64-row tiles and the threshold is the rank-th
  // largest of those maxima (launch_sample_tau_groups, the K2s sample's scheme: every maximum is one of the sample's
  // scores, so the threshold is at most the dense one -- a few rows more pass).  The radix select over 8 x 65 536
  // scores took 41 us alone and 0.4-0.7 ms beside the other context's sweep (a string of dependent trips to a
  // saturated memory: profiles/r05_funnel64_trace_excerpt.txt); this one reads 4 KB per query.  Where the rank is not
  // small against the number of tiles (small corpora: the sample is most of the rows) the dense form stays.
  const bool by_maxima = (uint64_t)rank * 16 <= stiles && stiles <= 2048;
  // one upload: full queries [nq][ld] (f32), their prefixes as f64 [8][ldq] (what stage 1 reads, through
  // the scalar cache; ld and ldq are multiples of 64, so the block stays 32-byte aligned), list lengths
  const bool cosine = ix->metric == VT_COSINE;
  static_assert(vt::kPrefixMultiMax == vt::kCosineMultiMax, "one group size");
  // (K1p reads all eight query rows whatever nq is: the buffer always holds eight)
  const size_t q_floats = (size_t)vt::kCosineMultiMax * ld, p_floats = (size_t)vt::kCosineMultiMax * ldq * 2;
  const size_t up_floats = q_floats + p_floats + 2 * vt::kCosineMultiMax;  // (+ [8] candidates per list, [8] keys per K1 stage list)
  VT_TRY(c.dBQ.ensure(up_floats));
  VT_TRY(c.hBQ.ensure(up_floats));
  VT_TRY(c.dBSample.ensure((size_t)vt::kCosineMultiMax * sample_rows));
  VT_TRY(c.dBTau.ensure(vt::kCosineMultiMax));
  VT_TRY(c.dBCount.ensure(vt::kCosineMultiMax));
  VT_TRY(c.dPartKeys.ensure((size_t)nq * kListCap));
  VT_TRY(c.dPartPay.ensure((size_t)nq * kListCap));
  VT_TRY(c.dStageB.ensure(nq));
  constexpr uint32_t kStageBlocks = 4;  // K1 batch mode: blocks per query over its <= 256 candidates (8-row tiles)
  VT_TRY(c.dCandKeys.ensure((size_t)nq * k1 * kStageBlocks));
  VT_TRY(c.dCandPay.ensure((size_t)nq * k1 * kStageBlocks));
  // (a list that came out short leaves the tail of its block as it was: rows a later stage may still gather --
  // zeroed, they are row 0)
  VT_HIP(hipMemsetAsync(c.dStageB.p, 0, (size_t)nq * sizeof(ResultBlock), c.stream));
  const size_t res_bytes = (size_t)vt::kHammingMultiMax * vt::kMaxFusedK * sizeof(vt::Entry) + 256;
  if (!c.dBigMapped || c.hBig.count < res_bytes) {
    VT_TRY(c.hBig.ensure(std::max<size_t>(res_bytes, 16 + (size_t)vt::kSelListMax * sizeof(vt::Entry))));
    VT_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&c.dBigMapped), c.hBig.p, 0));
  }
  const size_t ent_bytes = (size_t)nq * k2 * sizeof(vt::Entry);
  vt::Entry *dOut = reinterpret_cast<vt::Entry *>(c.dBigMapped);
  uint32_t *dOutCount = reinterpret_cast<uint32_t *>(c.dBigMapped + ent_bytes);
  const vt::Entry *hOut = reinterpret_cast<const vt::Entry *>(c.hBig.p);
  const uint32_t *hOutCount = reinterpret_cast<const uint32_t *>(c.hBig.p + ent_bytes);
  int *hStatus = reinterpret_cast<int *>(c.hBig.p + ent_bytes + 32);
  uint32_t *hListCount = reinterpret_cast<uint32_t *>(c.hBig.p + ent_bytes + 64);  // [8]: rows that reached tau
  float *hTau = reinterpret_cast<float *>(c.hBig.p + ent_bytes + 96);             // [8]: the thresholds themselves
  uint64_t *hLastKey = reinterpret_cast<uint64_t *>(c.hBig.p + ent_bytes + 128);  // [8]: key of each list's k1-th (last kept) row
  std::memset(c.hBQ.p, 0, up_floats * sizeof(float));
  vt::CosineScanMultiArgs a{};
  uint32_t *hcounts = reinterpret_cast<uint32_t *>(c.hBQ.p + q_floats + p_floats);
  for (uint32_t i = 0; i < nq; ++i) {
    const float *q = queries + which[i] * d;
    std::memcpy(c.hBQ.p + (size_t)i * ld, q, (size_t)d * sizeof(float));
    hcounts[i] = k1;
    hcounts[vt::kCosineMultiMax + i] = kStageBlocks * k1;
    if (!cosine) continue;
    double *qd = reinterpret_cast<double *>(c.hBQ.p + q_floats) + (size_t)i * ldq;
    double qq = 0.0;  // f64_dot(q, q) over the prefix (distances.rs:179-185)
    for (uint32_t j = 0; j < d1; ++j) {
      qd[j] = (double)q[j];
      qq += (double)q[j] * (double)q[j];
    }
    a.qq[i] = qq;
  }
  VT_HIP(hipMemcpyAsync(c.dBQ.p, c.hBQ.p, up_floats * sizeof(float), hipMemcpyHostToDevice, c.stream));
  const uint32_t *dcounts = reinterpret_cast<const uint32_t *>(c.dBQ.p + q_floats + p_floats);
  const uint32_t *dlens = dcounts + vt::kCosineMultiMax;
  VT_HIP(hipMemsetAsync(c.dBCount.p, 0, vt::kCosineMultiMax * sizeof(uint32_t), c.stream));
  a.X = ix->dX;
  a.stride = ix->ld;
  a.Qd = reinterpret_cast<const double *>(c.dBQ.p + q_floats);
  a.id_rank = ix->dRank.p;
  a.n = n;
  a.d = d1;
  a.nq = nq;
  a.status = c.dStatus.p;
  vt::PrefixMultiArgs pa{};
  pa.X = ix->dX;
  pa.stride = ix->ld;
  pa.Q = c.dBQ.p;
  pa.q_stride = ld;
  pa.id_rank = ix->dRank.p;
  pa.n = n;
  pa.d = d1;
  pa.nq = nq;
  pa.metric = ix->metric;
  pa.order = ix->order;
  pa.status = c.dStatus.p;
  const size_t lds = cosine ? vt::cosine_scan_multi_lds_bytes() : vt::prefix_multi_lds_bytes();
  // pass 0: the sample's scores -> one threshold per query
  a.sample = pa.sample = c.dBSample.p;
  a.sample_stride = pa.sample_stride = sstride;
  a.sample_rows = pa.sample_rows = by_maxima ? stiles : sample_rows;
  a.sample_maxima = pa.sample_maxima = by_maxima ? 1u : 0u;
  if (cosine) VT_HIP(vt::launch_cosine_scan_multi(a, c.grid_for(stiles, lds), c.stream));
  else VT_HIP(vt::launch_prefix_multi(pa, c.grid_for(stiles, lds, vt::prefix_multi_blocks_per_cu()), c.stream));
  if (by_maxima) VT_HIP(vt::launch_sample_tau_groups(c.dBSample.p, stiles, vt::kCosineMultiMax, nq, rank, c.dBTau.p, c.stream));
  else VT_HIP(vt::launch_sample_tau(c.dBSample.p, sample_rows, vt::kCosineMultiMax, nq, rank, c.dBTau.p, c.stream));
  // pass 1: every row's prefix once; (query, row) pairs at or above the thresholds into the lists
  a.sample = pa.sample = nullptr;
  a.sample_maxima = pa.sample_maxima = 0;
  a.tau = pa.tau = c.dBTau.p;
  a.cand_keys = pa.cand_keys = c.dPartKeys.p;
  a.cand_pay = pa.cand_pay = c.dPartPay.p;
  a.cand_count = pa.cand_count = c.dBCount.p;
  a.cand_cap = pa.cand_cap = kListCap;
  if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
  if (cosine) VT_HIP(vt::launch_cosine_scan_multi(a, c.grid_for(ntiles, lds), c.stream));
  else VT_HIP(vt::launch_prefix_multi(pa, c.grid_for(ntiles, lds, vt::prefix_multi_blocks_per_cu()), c.stream));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
  VT_HIP(hipMemcpyAsync(hListCount, c.dBCount.p, vt::kCosineMultiMax * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(vt::launch_select_lists(c.dPartKeys.p, c.dPartPay.p, nq, kListCap, c.dBCount.p, k1, c.dStageB.p,
                                 (uint32_t)sizeof(ResultBlock), c.stream, /*sixteen blocks per list=*/true));
  // what the acceptance test below looks at: the thresholds and the key of every list's last kept row
  // (copied out here: later stages reuse the blocks)
  VT_HIP(hipMemcpyAsync(hTau, c.dBTau.p, vt::kCosineMultiMax * sizeof(float), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipMemcpy2DAsync(hLastKey, sizeof(uint64_t), &c.dStageB.p->e[k1 - 1].key, sizeof(ResultBlock), sizeof(uint64_t), nq,
                          hipMemcpyDeviceToHost, c.stream));
  // later stages re-score the same candidates on a longer prefix (collection.ex:674-691), then
  // exact_rerank on the full vectors (collection.ex:821-851): the queries on grid.y
  vt::CosineRerankArgs r{};
  r.X = ix->dX;
  r.stride = ix->ld;
  r.q = c.dBQ.p;
  r.id_rank = ix->dRank.p;
  r.gather = &c.dStageB.p->e[0].row;
  r.gather_stride = sizeof(vt::Entry) / sizeof(uint32_t);
  r.n = k1;
  r.out_keys = c.dCandKeys.p;
  r.out_pay = c.dCandPay.p;
  r.status = c.dStatus.p;
  r.q_stride = ld;
  r.gather_qstride = (uint32_t)(sizeof(ResultBlock) / sizeof(uint32_t));
  // (the other families: K1 over each query's candidate rows, kStageBlocks lists of `k` per query)
  vt::ScanArgs sa{};
  sa.X = ix->dX;
  sa.stride = ix->ld;
  sa.q = c.dBQ.p;
  sa.id_rank = ix->dRank.p;
  sa.gather = &c.dStageB.p->e[0].row;
  sa.gather_stride = sizeof(vt::Entry) / sizeof(uint32_t);
  sa.n = k1;
  sa.metric = ix->metric;
  sa.order = ix->order;
  sa.part_keys = c.dCandKeys.p;
  sa.part_pay = c.dCandPay.p;
  sa.status = c.dStatus.p;
  sa.batch_counts = dcounts;
  sa.batch_cap = k1;
  sa.batch_qstride = ld;
  sa.batch_gather_stride = (uint32_t)(sizeof(ResultBlock) / sizeof(uint32_t));
  static const uint32_t kListsPerQuery = vt::scan_lists(kStageBlocks);
  for (size_t i = 1; i < nstages; ++i) {
    if (cosine) {
      r.d = (uint32_t)stages[i];
      VT_HIP(vt::launch_cosine_rerank_batch(r, nq, c.stream));
      VT_HIP(vt::launch_select_lists(c.dCandKeys.p, c.dCandPay.p, nq, k1, dcounts, k1, c.dStageB.p, (uint32_t)sizeof(ResultBlock),
                                     c.stream));
    } else {
      sa.d = (uint32_t)stages[i];
      sa.k = k1;
      VT_HIP(vt::launch_scan_batch(sa, kStageBlocks, nq, c.stream));
      VT_HIP(vt::launch_select_lists(c.dCandKeys.p, c.dCandPay.p, nq, kListsPerQuery * k1, dlens, k1, c.dStageB.p,
                                     (uint32_t)sizeof(ResultBlock), c.stream));
    }
  }
  if (cosine) {
    r.d = d;
    VT_HIP(vt::launch_cosine_rerank_batch(r, nq, c.stream));
    VT_HIP(vt::launch_batch_select(c.dCandKeys.p, c.dCandPay.p, nq, k1, k2, dOut, dOutCount, c.stream));
  } else {
    sa.d = d;
    sa.k = k2;
    VT_HIP(vt::launch_scan_batch(sa, kStageBlocks, nq, c.stream));
    VT_HIP(vt::launch_batch_select(c.dCandKeys.p, c.dCandPay.p, nq, kListsPerQuery * k2, k2, dOut, dOutCount, c.stream));
  }
  VT_HIP(hipMemcpyAsync(hStatus, c.dStatus.p, sizeof(int), hipMemcpyDeviceToHost, c.stream));  // (pinned: stays asynchronous)
  VT_HIP(hipMemsetAsync(c.dStatus.p, 0, sizeof(int), c.stream));
  run.which = which;
  run.nq = nq;
  run.k1 = k1;
  run.k2 = k2;
  run.n = n;
  run.d1 = d1;
  run.cosine = cosine;
  run.as_scan = as_scan;
  run.hOut = hOut;
  run.hOutCount = hOutCount;
  run.hStatus = hStatus;
  run.hListCount = hListCount;
  run.hTau = hTau;
  run.hLastKey = hLastKey;
  return VT_OK;
}

int funnel_group_finish(Shard *ix, Ctx &c, const FunnelGroupRun &run, vt_hits **out, std::vector<char> &done) {
  const std::vector<size_t> &which = run.which;
  const uint32_t nq = run.nq, k1 = run.k1, k2 = run.k2, n = run.n, d1 = run.d1;
  const bool cosine = run.cosine, as_scan = run.as_scan;
  const vt::Entry *hOut = run.hOut;
  const uint32_t *hOutCount = run.hOutCount, *hListCount = run.hListCount;
  const int *hStatus = run.hStatus;
  const float *hTau = run.hTau;
  const uint64_t *hLastKey = run.hLastKey;
  constexpr uint32_t kListCap = 8192;
  VT_HIP(hipStreamSynchronize(c.stream));
  if (c.profiling) {
    float ms = 0.f;
    VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    if (as_scan) {  // (a batch of plain searches: the sweep is its scan)
      c.prof.scan_launches += 1;
      c.prof.scan_ms += ms;
      c.prof.scan_rows += n;
      c.prof.scan_bytes += (uint64_t)n * d1 * 4;
      c.prof.sweep_queries += nq;
    } else {
      c.prof.prefix_launches += 1;
      c.prof.prefix_ms += ms;
      c.prof.prefix_bytes += (uint64_t)n * d1 * 4;
      c.prof.prefix_queries += nq;
    }
  }
  if (*hStatus != 0) return kRetryInternal;  // an overflow somewhere: one by one, each reports its own
  auto orderable_host = [](float f) {  // f32::total_cmp as an order-preserving u32 (the device's `orderable`)
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  };
  for (uint32_t i = 0; i < nq; ++i) {
    if (hListCount[i] < k1 || hListCount[i] > kListCap) continue;  // the threshold missed: this one takes the single path
    // The list was cut by raw >= tau, the single path cuts by key = orderable(1.0f - raw) << 32 | id rank: below
    // raw = 0.5 the f32 subtraction puts neighbouring raws on ONE rank, so a row just under tau can carry the rank of
    // the list's last kept row and beat it on its id -- the single path would keep it, the list never saw it
    // (ADVICE r3).  Every excluded row has 1.0f - raw >= 1.0f - tau (rounding is monotone): the list is the single
    // path's exactly when its last kept rank lies strictly below the rank of tau itself.
    // (the other families cut by the order their keys sort in: nothing to check)
    if (cosine && (uint32_t)(hLastKey[i] >> 32) >= orderable_host(1.0f - hTau[i])) continue;
    const uint32_t got = std::min<uint32_t>(hOutCount[i], k2);
    std::vector<vt::Entry> entries(hOut + (size_t)i * k2, hOut + (size_t)i * k2 + got);
    VT_TRY(make_hits(ix, entries, &out[which[i]]));
    done[which[i]] = 1;
  }
  return VT_OK;
}

// One group, start to end, on one context.
int funnel_group(Shard *ix, Ctx &c, const float *queries, const std::vector<size_t> &which, const size_t *stages,
                 size_t nstages, size_t candidates, size_t limit, vt_hits **out, std::vector<char> &done, bool as_scan) {
  FunnelGroupRun run;
  const int st = funnel_group_queue(ix, c, run, queries, which, stages, nstages, candidates, limit, as_scan);
  if (st != VT_OK) {  // (whatever was queued reads the context's pinned blocks)
    const std::string why = g_last_error;
    (void)hipStreamSynchronize(c.stream);
    (void)hipGetLastError();
    g_last_error = why;
    return st;
  }
  return funnel_group_finish(ix, c, run, out, done);
}

// The groups of one call (r05).  A group is a sweep of the corpus for up to eight queries (0.9 ms over a 128-float
// prefix, 4.7-5.7 ms over whole rows) and then a tail nobody else can use the card under: list selects, the later
// stages' rescoring of 8 x `candidates` rows, the exact rerank, a select, the wait, eight hit lists -- 0.3-0.5 ms,
// a third of a prefix-128 group (a batch of 64 cosine funnels: 11.0 ms = 8 x 1.375 for 8 x 0.89 ms of sweeps).  With a
// second context the groups alternate: group g + 1 is staged and queued BEFORE group g is waited for, its sample pass
// and sweep start while g's tail kernels run, g's host side runs under g + 1's sweep.  Each group's lists are what it
// gives alone: nothing is shared but the card.  A group that reports kRetryInternal (an overflow somewhere in it) hands
// its queries back undone; any other failure ends the call once both streams are idle.
int funnel_groups(Shard *ix, Ctx &c, const float *queries, const std::vector<std::vector<size_t>> &groups, const size_t *stages,
                  size_t nstages, size_t candidates, size_t limit, vt_hits **out, std::vector<char> &done, bool as_scan) {
  auto settle = [&](const std::vector<size_t> &which, int st) -> int {
    if (st != kRetryInternal) return st;
    for (size_t i : which) {
      delete out[i];
      out[i] = nullptr;
      done[i] = 0;
    }
    return VT_OK;
  };
  SpareCtxLease spare(groups.size() >= 2 && !vt::env::on(vt::env::NO_GROUP_PIPELINE) ? ix : nullptr);
  if (!spare.c) {
    for (const auto &which : groups) VT_TRY(settle(which, funnel_group(ix, c, queries, which, stages, nstages, candidates, limit, out, done, as_scan)));
    return VT_OK;
  }
  VT_TRY(reader_ready(ix, *spare.c));
  Ctx *cx[2] = {&c, spare.c};
  FunnelGroupRun runs[2];
  auto queue = [&](size_t g) {
    return funnel_group_queue(ix, *cx[g & 1], runs[g & 1], queries, groups[g], stages, nstages, candidates, limit, as_scan);
  };
  int st = queue(0);
  for (size_t g = 0; g < groups.size() && st == VT_OK; ++g) {
    const int st_next = g + 1 < groups.size() ? queue(g + 1) : VT_OK;
    st = settle(groups[g], funnel_group_finish(ix, *cx[g & 1], runs[g & 1], out, done));
    if (st == VT_OK) st = st_next;
  }
  if (st != VT_OK) {  // (whatever is still queued reads and writes the two contexts' pinned blocks)
    const std::string why = g_last_error;
    (void)hipStreamSynchronize(cx[0]->stream);
    (void)hipStreamSynchronize(cx[1]->stream);
    (void)hipGetLastError();
    g_last_error = why;
  }
  return st;
}

// funnel_search for nq queries (rows of `queries`) with one set of stages: groups of up to eight
// share the stage-1 sweep; what the groups cannot take goes through funnel_ready one by one.
int funnel_batch_ready(Shard *ix, Ctx &c, const float *queries, size_t nq, size_t d, const size_t *stages, size_t nstages,
                       size_t candidates, size_t limit, vt_hits **out) {
  for (size_t i = 0; i < nq; ++i) VT_TRY(validate_vector(queries + i * d, d, ix->dim));
  if (nstages == 0) return VT_ERR_PREFIX;
  for (size_t i = 0; i < nstages; ++i)
    if (stages[i] == 0 || stages[i] > d) return VT_ERR_PREFIX;
  if (ix->n == 0 || candidates == 0 || limit == 0) {
    for (size_t i = 0; i < nq; ++i) VT_TRY(empty_hits(&out[i]));
    return VT_OK;
  }
  std::vector<char> done(nq, 0);
  if (nq >= 2 && funnel_group_applies(ix, stages, nstages, candidates, limit)) {
    std::vector<std::vector<size_t>> groups;
    for (size_t g0 = 0; g0 < nq; g0 += vt::kCosineMultiMax) {
      std::vector<size_t> which;
      for (size_t i = g0; i < std::min<size_t>(nq, g0 + vt::kCosineMultiMax); ++i) which.push_back(i);
      if (which.size() < 2) break;
      groups.push_back(std::move(which));
    }
    VT_TRY(funnel_groups(ix, c, queries, groups, stages, nstages, candidates, limit, out, done, false));
  }
  for (size_t i = 0; i < nq; ++i)
    if (!done[i]) VT_TRY(funnel_ready(ix, c, queries + i * d, d, stages, nstages, candidates, limit, &out[i]));
  return VT_OK;
}

}  // namespace
