// vt_batch.h -- query batches on one shard: the matrix-core groups (K2 / K2b / K2s + exact rescoring, pipelined over two contexts), K1m / K4p / K1p sweeps, batch_ready
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// ---------------------------------------------------------------- K2 / K2b host side
// One group of <= 256 queries through the matrix cores.  `done[i]` is set for
// every query whose exact top-k was proven complete; the others are left for
// the single-query path.
//
// bf16 = false: K2, operands in f32 on v_mfma_f32_32x32x2_f32 (MFMA-bound).
// bf16 = true:  K2b, operands rounded to bf16 on v_mfma_f32_32x32x16_bf16 (HBM-bound); the
//               acceptance margin grows by the operand rounding, nothing else changes.
// tau_given (K2b's second pass): thresholds known from a first pass's exact results instead of
// a sample.  retry_tau (first pass only): for every query the bound could not certify but whose
// k exact hits exist, the threshold with which a second pass is certain to certify it (NaN
// where there is none).
//
// Two phases, so that consecutive groups of one call overlap (batch_ready): batch_group_queue stages the
// queries and queues every kernel and copy of the group on the context's stream without waiting for any of
// it; batch_group_finish waits for that stream, judges every query and builds the accepted ones' hit lists.
// Between the two the context belongs to the group (its pinned blocks are the group's upload and results).
struct BatchGroupRun {
  const float *queries = nullptr;
  size_t nq = 0, limit = 0;
  bool bf16 = false, shadow = false, tau_given = false;
  uint32_t k = 0, nq_pad = 0, cand_cap = 0;
  std::vector<double> qnorm;
  std::chrono::steady_clock::time_point t_begin;
  double t_staged = 0, t_queued = 0;
};

// `idle_cus` (consecutive groups of one call, batch_ready): the pass over the rows leaves that many CUs without a block
// of its own.  K2s keeps one block per CU resident from the first row to the last (all of the CU's LDS), so whatever
// else is queued meanwhile -- the previous group's exact rescoring and select, the next group's sample pass and
// thresholds -- runs between two passes unless some CUs are left for it.
int batch_group_queue(Shard *ix, Ctx &c, BatchGroupRun &run, const float *queries, size_t nq, size_t limit, bool bf16,
                      const float *tau_given, uint32_t idle_cus = 0, bool pipelined = false) {
  const uint32_t d = (uint32_t)ix->dim, ld = ix->ld, n = ix->n;
  const uint32_t k = (uint32_t)std::min<size_t>(limit, n);
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
  uint32_t nq_pad = bf16 ? vt::batch_bf16_pad((uint32_t)nq) : 32;
  while (nq_pad < nq) nq_pad *= 2;
  // K2s: the pass reads the bf16 shadow of the rows when the shard keeps a current one (half the bytes, no
  // conversion; the same rounding, so everything below -- sample, tau, bound -- is K2b's)
  const bool shadow = bf16 && shadow_wanted(ix) && shadow_current(ix);
  const uint32_t rows_per_block = shadow ? vt::batch_shadow_rows_per_block()
                                  : bf16 ? vt::batch_bf16_rows_per_block()
                                         : vt::batch_rows_per_block(nq_pad);
  const uint32_t ntiles_total = (n + rows_per_block - 1) / rows_per_block;
  // pass-0 sample: 1/64 of the row tiles, 128..512 of them, spread over the corpus.
  // A larger sample gives a tighter tau: fewer candidates to rescore and, above
  // all, fewer trips through the epilogue's append path (a returning global
  // atomic, ~2 us with the matrix pipe idle: 5 % of the pass at 128 tiles).
  // (at most 65 536 sample rows where the whole sample matrix is kept: sample_tau_kernel holds a query's sample in
  // registers.  K2s files one value per query and 64-row GROUP instead -- the group's best score -- and takes the
  // threshold from those, as long as the rank is small beside the groups (two of the `rank` best rarely share one);
  // it may then sample more tiles for the same money: VT_BATCH_SAMPLE_TILES, A/B.)
  struct SamplePlan {
    uint32_t stride, ntiles, rows, rank, groups;
    bool by_maxima;
  };
  auto plan_sample = [&](uint32_t tiles_cap) {
    SamplePlan sp{};
    const uint32_t want_tiles = std::min<uint32_t>(std::min<uint32_t>(512, tiles_cap), std::max<uint32_t>(128, ntiles_total / 64));
    sp.stride = std::max<uint32_t>(1, (ntiles_total + want_tiles - 1) / want_tiles);
    sp.ntiles = (ntiles_total + sp.stride - 1) / sp.stride;
    sp.rows = sp.ntiles * rows_per_block;
    // tau = rank-th best sample score: about rank * n / sample_rows rows pass.  K2b's margin is
    // ~0.2 sigma of a score distribution where K2's is ~1e-4, so its tau sits lower: the k-th hit
    // must clear it by that margin or the query costs a second pass.
    const double ratio = (double)sp.rows / (double)n;
    const double want_cand = bf16 ? std::min(32.0 * k, std::max(8.0 * k, 4096.0)) : 8.0 * k;
    uint32_t rank = (uint32_t)std::ceil(want_cand * std::min(1.0, ratio));
    rank = std::max<uint32_t>(bf16 ? bf16_min_rank() : 3, std::min<uint32_t>(rank, std::min<uint32_t>(sp.rows, n)));
    // test hook (vt_debug_set "bf16_rank"): K2b's threshold from exactly the r-th best sample score
    // (r = limit leaves no margin at all: every query then needs the second pass)
    if (bf16) {
      const long v = vt::env::get(vt::env::BF16_RANK);
      if (v >= 1) rank = std::min<uint32_t>((uint32_t)v, std::min<uint32_t>(sp.rows, n));
    }
    sp.rank = rank;
    sp.groups = vt::batch_shadow_sample_groups(sp.ntiles);
    sp.by_maxima = shadow && !tau_given && (uint64_t)rank * 16 <= sp.groups && sp.groups <= 2048;
    return sp;
  };
  const uint32_t dense_cap = 65536 / rows_per_block;
  // (K2s: 512 tiles -- 33 us -> 60 us of sample pass, half the candidates to rescore behind the pass: A.15)
  SamplePlan sp = plan_sample(!shadow ? dense_cap : 512u);
  if (!sp.by_maxima && sp.rows > 65536) sp = plan_sample(dense_cap);
  const uint32_t stride = sp.stride, ntiles_sample = sp.ntiles, sample_rows = sp.rows, rank = sp.rank, sample_groups = sp.groups;
  const bool by_maxima = sp.by_maxima;
  const uint32_t cand_cap = 8192;
  constexpr uint32_t kBlocksPerQuery = 8;  // blocks of the exact rescoring per query (2 / 4 / 8 / 16: 339 / 241 / 171 / 168 us per 256 queries)

  VT_TRY(c.dBQ.ensure((size_t)nq_pad * ld));
  VT_TRY(c.hBQ.ensure((size_t)nq_pad * ld));
  VT_TRY(c.dBTau.ensure(nq_pad));
  VT_TRY(c.hBTau.ensure(nq_pad));
  if (!tau_given) VT_TRY(c.dBSample.ensure((size_t)nq_pad * (by_maxima ? sample_groups : sample_rows)));
  VT_TRY(c.dBCand.ensure((size_t)nq_pad * cand_cap));
  VT_TRY(c.dBCount.ensure(nq_pad));
  VT_TRY(c.hBCount.ensure(nq_pad));
  VT_TRY(c.hBOut.ensure((size_t)nq_pad * k));
  VT_TRY(c.hBOutCount.ensure(nq_pad + 1));  // (+ the status word: a copy into pageable memory would wait for the stream)
  VT_TRY(c.dPartKeys.ensure((size_t)nq_pad * kBlocksPerQuery * k));
  VT_TRY(c.dPartPay.ensure((size_t)nq_pad * kBlocksPerQuery * k));
  if (bf16) VT_TRY(c.dBQimage.ensure(std::max(vt::batch_bf16_image_bytes(ld), vt::batch_shadow_image_bytes(ld))));

  run.queries = queries;
  run.nq = nq;
  run.limit = limit;
  run.bf16 = bf16;
  run.shadow = shadow;
  run.tau_given = tau_given != nullptr;
  run.k = k;
  run.nq_pad = nq_pad;
  run.cand_cap = cand_cap;
  run.t_begin = t_begin;
  std::vector<double> &qnorm = run.qnorm;  // (filled while the device works: see below)
  qnorm.assign(nq, 0.0);
  std::memset(c.hBQ.p, 0, (size_t)nq_pad * ld * sizeof(float));
  for (size_t i = 0; i < nq; ++i) std::memcpy(c.hBQ.p + i * ld, queries + i * d, (size_t)d * sizeof(float));
  run.t_staged = since();
  VT_HIP(hipMemcpyAsync(c.dBQ.p, c.hBQ.p, (size_t)nq_pad * ld * sizeof(float), hipMemcpyHostToDevice, c.stream));
  vt::BatchScoreArgs a{};
  a.X = ix->dX;
  a.stride = ix->ld;
  a.Q = c.dBQ.p;
  a.ld = ld;
  a.nq_pad = nq_pad;
  a.n_total = n;
  const bool l2_family = ix->metric == VT_L2 || ix->metric == VT_L2_SQUARED;
  a.xnorm2 = l2_family ? ix->dXnorm2.p : nullptr;
  if (bf16) {
    a.Qimage = c.dBQimage.p;
    if (shadow) {
      a.Xshadow = ix->dShadow.p;
      VT_HIP(vt::launch_batch_q_image16(c.dBQ.p, ld, nq_pad, c.dBQimage.p, c.stream));
    } else {
      VT_HIP(vt::launch_batch_q_image(c.dBQ.p, ld, nq_pad, c.dBQimage.p, c.stream));
    }
  }
  auto scores = [&](bool dense, uint32_t blocks) {
    if (shadow) return vt::launch_batch_scores_shadow(a, dense, blocks, c.stream);
    return bf16 ? vt::launch_batch_scores_bf16(a, dense, blocks, c.stream) : vt::launch_batch_scores(a, dense, blocks, c.stream);
  };
  const uint32_t grid_cap = (uint32_t)c.num_cus;
  if (tau_given) {
    for (uint32_t i = 0; i < nq_pad; ++i) c.hBTau.p[i] = i < nq ? tau_given[i] : INFINITY;
    VT_HIP(hipMemcpyAsync(c.dBTau.p, c.hBTau.p, (size_t)nq_pad * sizeof(float), hipMemcpyHostToDevice, c.stream));
  } else {
    // pass 0: dense scores of the sample -> tau
    a.n = sample_rows;
    a.sample_stride = stride;
    a.sample = c.dBSample.p;
    a.sample_rows = sample_rows;
    if (by_maxima) {
      a.sample_rows = sample_groups;
      VT_HIP(vt::launch_batch_sample_maxima_shadow(a, std::min<uint32_t>(ntiles_sample, grid_cap), c.stream));
      VT_HIP(vt::launch_sample_tau_groups(c.dBSample.p, sample_groups, nq_pad, (uint32_t)nq, rank, c.dBTau.p, c.stream));
    } else {
      VT_HIP(scores(true, std::min<uint32_t>(ntiles_sample, grid_cap)));
      VT_HIP(vt::launch_sample_tau(c.dBSample.p, sample_rows, nq_pad, (uint32_t)nq, rank, c.dBTau.p, c.stream));
    }
  }
  // pass 1: all rows, candidates with score >= tau
  a.n = n;
  a.sample = nullptr;
  a.tau = c.dBTau.p;
  a.cand = c.dBCand.p;
  a.cand_count = c.dBCount.p;
  a.cand_cap = cand_cap;
  VT_HIP(hipMemsetAsync(c.dBCount.p, 0, (size_t)nq_pad * sizeof(uint32_t), c.stream));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev2, c.stream));
  // (between groups of one call the pass keeps a four-stage ring: 32 KB of every CU's LDS stay free, so the neighbours'
  // small kernels -- the select behind the previous group's rescoring above all, 640 B of LDS and 26 registers -- are
  // dispatched beside the resident blocks of the pass instead of behind its last one: the trace of the five-stage
  // form showed batch_select_kernel waiting 3.6 ms for a CU, and with it the host and the group after next)
  if (shadow && pipelined) a.stages = 4;
  VT_HIP(scores(false, std::min<uint32_t>(ntiles_total, shadow && idle_cus < grid_cap / 2 ? grid_cap - idle_cus : grid_cap)));
  a.stages = 0;
  if (c.profiling) VT_HIP(hipEventRecord(c.ev3, c.stream));
  // exact rescoring of every query's candidates with the K1 arithmetic
  vt::ScanArgs sa{};
  sa.X = ix->dX;
  sa.stride = ix->ld;
  sa.q = c.dBQ.p;
  sa.id_rank = ix->dRank.p;
  sa.gather = &c.dBCand.p->row;
  sa.gather_stride = sizeof(vt::BatchCand) / sizeof(uint32_t);
  sa.n = cand_cap;
  sa.d = d;
  sa.metric = ix->metric;
  sa.order = ix->order;
  sa.k = k;
  sa.part_keys = c.dPartKeys.p;
  sa.part_pay = c.dPartPay.p;
  sa.status = c.dStatus.p;
  sa.batch_counts = c.dBCount.p;
  sa.batch_cap = cand_cap;
  VT_HIP(vt::launch_scan_batch(sa, kBlocksPerQuery, nq_pad, c.stream));
  // the lists, their counts, every query's candidate count and threshold and the status word leave with the select
  // kernel, written straight into the host-mapped blocks (r05: four blit launches and a memset per group before)
  vt::BatchExport ex{};
  ex.cand_count = c.dBCount.p;
  ex.cand_count_out = c.hBCount.mapped();
  ex.tau = tau_given ? nullptr : c.dBTau.p;
  ex.tau_out = tau_given ? nullptr : c.hBTau.mapped();
  ex.status = c.dStatus.p;
  ex.status_out = reinterpret_cast<int *>(c.hBOutCount.mapped() + nq_pad);
  if (!c.hBOut.mapped() || !c.hBOutCount.mapped() || !ex.cand_count_out || (!tau_given && !ex.tau_out))
    return fail(VT_ERR_DEVICE, "hipHostGetDevicePointer (batch results)");
  VT_HIP(vt::launch_batch_select(c.dPartKeys.p, c.dPartPay.p, nq_pad, kBlocksPerQuery * k, k, c.hBOut.mapped(), c.hBOutCount.mapped(),
                                 c.stream, &ex));
  run.t_queued = since();
  // the queries' norms (the acceptance bound needs them): 0.1 ms of host work per 256 x 768, done
  // while the device runs its 4 ms
  for (size_t i = 0; i < nq; ++i) {
    double s = 0.0;
    for (uint32_t j = 0; j < d; ++j) s += (double)queries[i * d + j] * (double)queries[i * d + j];
    qnorm[i] = std::sqrt(s);
  }
  return VT_OK;
}

// `done[i]` is set for every query of the group whose exact top-k was proven complete (its hits are in out[i]).
int batch_group_finish(Shard *ix, Ctx &c, BatchGroupRun &run, vt_hits **out, std::vector<char> &done, std::vector<float> *retry_tau) {
  const uint32_t d = (uint32_t)ix->dim, ld = ix->ld, n = ix->n;
  const size_t nq = run.nq;
  const bool bf16 = run.bf16, shadow = run.shadow, tau_given = run.tau_given;
  const uint32_t k = run.k, nq_pad = run.nq_pad, cand_cap = run.cand_cap;
  const std::vector<double> &qnorm = run.qnorm;
  const bool l2_family = ix->metric == VT_L2 || ix->metric == VT_L2_SQUARED;
  const bool trace = vt::env::on(vt::env::TRACE_BATCH);  // phases of a group on stderr
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - run.t_begin).count(); };
  VT_HIP(hipStreamSynchronize(c.stream));
  const double t_synced = since();
  const int status = (int)c.hBOutCount.p[nq_pad];
  if (c.profiling) {
    float ms = 0.f;
    VT_HIP(hipEventElapsedTime(&ms, c.ev2, c.ev3));
    if (bf16) {
      c.prof.nominate_launches += 1;
      c.prof.nominate_ms += ms;
      c.prof.nominate_bytes += (uint64_t)n * d * (shadow ? 2 : 4);
      c.prof.nominate_shadow_launches += shadow ? 1 : 0;
      c.prof.nominate_flops += 2.0 * (double)n * (double)nq_pad * (double)ld;
      c.prof.nominate_queries += nq;
      c.prof.nominate_second_passes += tau_given ? 1 : 0;
      uint64_t cands = 0;
      for (size_t i = 0; i < nq; ++i) cands += std::min<uint32_t>(c.hBCount.p[i], cand_cap);
      c.prof.nominate_candidates += cands;
    } else {
      c.prof.batch_launches += 1;
      c.prof.batch_ms += ms;
      c.prof.batch_flops += 2.0 * (double)n * (double)nq_pad * (double)ld;
      c.prof.batch_queries += nq;
    }
  }
  if (status != 0) return VT_OK;  // an exact rescoring overflowed somewhere: let the single-query path decide

  // A query is accepted when no row outside its candidate set can reach the
  // top k.  Every such row y has score_mfma(y) < tau.  With u = 2^-24 and X the
  // largest row norm, both the MFMA sum and the reference's chunked sum are
  // d-term f32 sums of the same products, so
  //   dot family:  |dot_mfma - dot_ref| <= 2 gamma_d |q| X            =: eps
  //                => dot_ref(y) < tau + eps; accepted if tau + eps (+ slack) <= dot_k;
  //   L2 family:   score = 2 q.x - |x|^2 = |q|^2 - |q - x|^2, so
  //                l2sq_ref(y) > |q|^2 - tau - eps with eps = 3.5 d u (|q| + X)^2;
  //                accepted if l2sq_k (+ slack) <= |q|^2 - tau - eps.
  // The slack keeps y strictly behind the k-th hit even after the f32 rank
  // (1 - raw for cosine, sqrt for L2) collapses nearby values onto equal keys,
  // where the id tie-break could otherwise let y in.
  //
  // K2b: each operand is first rounded to bf16 (8 significant bits, round to nearest even:
  // relative error <= ub = 2^-8 each), so every product carries a relative error of at most
  // 2 ub + ub^2 and, by Cauchy-Schwarz, the sum an absolute one of at most (2 ub + ub^2) |q| X;
  // the bf16 products are exact in f32 and the matrix core's f32 accumulation of them is priced
  // generously (8 d u |q| X, together with the reference's own summation error: whatever the
  // order and the rounding of its partial sums).  Subnormal operands may be flushed: at most
  // 2^-126 per element times the other operand, sqrt(d) 2^-126 (|q| + X) over a row.  Rounding
  // to bf16 can overflow only near f32's largest values: such queries are never certified here
  // (the guard below), nor is any whose margin is not finite.
  const double u = std::ldexp(1.0, -24), ub = std::ldexp(1.0, -8);
  const double xnorm = std::sqrt(ix->max_sqnorm);
  if (retry_tau) retry_tau->assign(nq, std::numeric_limits<float>::quiet_NaN());
  std::vector<size_t> accepted;
  accepted.reserve(nq);
  // (the winners' ids are 2 560 random picks from a table of millions: asked for ahead of the
  // loop that copies them, they arrive together -- 0.39 ms -> 0.1 ms per 256 queries)
  for (size_t i = 0; i < nq; ++i)
    for (uint32_t j = 0; j < std::min<uint32_t>(c.hBOutCount.p[i], k); ++j) {
      const uint32_t row = c.hBOut.p[i * k + j].row;
      if (row < ix->ids.size()) __builtin_prefetch(&ix->ids[row]);
    }
  for (size_t i = 0; i < nq; ++i) {
    const uint32_t cnt = c.hBCount.p[i];
    if (c.hBOutCount.p[i] < k) continue;
    const vt::Entry *e = c.hBOut.p + i * k;
    const double tau = (double)c.hBTau.p[i];
    const double raw_k = (double)e[k - 1].raw;
    const bool magnitudes_ok = !bf16 || (qnorm[i] < 1e18 && xnorm < 1e18);
    bool accept = false;
    double tau2 = std::numeric_limits<double>::quiet_NaN();  // threshold that certifies given these k exact hits
    if (l2_family) {
      double eps = 3.5 * (double)d * u * (qnorm[i] + xnorm) * (qnorm[i] + xnorm);
      if (bf16)
        eps = 2.0 * (2.0 * ub + ub * ub) * qnorm[i] * xnorm + 8.0 * (double)d * u * (qnorm[i] + xnorm) * (qnorm[i] + xnorm) +
              std::ldexp(1.0, -120) * std::sqrt((double)d) * (qnorm[i] + xnorm);
      const double l2sq_k = (ix->metric == VT_L2 ? raw_k * raw_k : raw_k) * (1.0 + 16.0 * u);
      const double bound = qnorm[i] * qnorm[i] * (1.0 - 4.0 * u) - eps;
      accept = l2sq_k <= bound - tau;
      tau2 = bound - l2sq_k * (1.0 + 16.0 * u);
    } else {
      double eps = 2.5 * (double)d * u * qnorm[i] * xnorm;
      if (bf16)
        eps = (2.0 * ub + ub * ub + 8.0 * (double)d * u) * qnorm[i] * xnorm +
              std::ldexp(1.0, -120) * std::sqrt((double)d) * (qnorm[i] + xnorm);
      const double dot_k = ix->metric == VT_NEG_INNER_PRODUCT ? -raw_k : raw_k;
      const double slack = ix->metric == VT_COSINE ? 4.0 * u * std::max(1.0, std::fabs(1.0 - dot_k)) : 0.0;
      accept = tau + eps + slack <= dot_k;
      tau2 = dot_k - eps - 2.0 * slack - 16.0 * u * std::fabs(dot_k);
    }
    accept = accept && magnitudes_ok && cnt <= cand_cap;
    if (!accept) {  // also taken when anything above is NaN
      if (retry_tau && magnitudes_ok && std::isfinite(tau2)) {
        // rounded DOWN to f32: the second pass nominates at least what tau2 asks for
        float t = (float)tau2;
        if ((double)t > tau2) t = std::nextafterf(t, -INFINITY);
        // (only a lower bar than the one that failed can help, and only a list that did not overflow)
        if (!tau_given && (double)t < tau && cnt <= cand_cap) (*retry_tau)[i] = t;
      }
      continue;
    }
    accepted.push_back(i);
  }
  // (the hit lists on four threads instead of this one: measured, 0.13 ms either way -- 256 lists of ten short ids are
  // four small allocations each, and three thread starts cost what they save)
  for (size_t i : accepted) {
    const vt::Entry *e = c.hBOut.p + i * k;
    std::vector<vt::Entry> entries(e, e + k);
    VT_TRY(make_hits(ix, entries, &out[i]));
    done[i] = 1;
  }
  if (trace)
    std::fprintf(stderr, "[vt] batch group nq=%zu %s: staged %.3f ms, queued %.3f, device done %.3f, hits built %.3f\n", nq,
                 bf16 ? "bf16" : "f32", run.t_staged, run.t_queued, t_synced, since());
  return VT_OK;
}

constexpr uint32_t kBatchTailCus = 0;

// One group, start to end, on one context.
int batch_group(Shard *ix, Ctx &c, const float *queries, size_t nq, size_t limit, vt_hits **out, std::vector<char> &done,
                bool bf16, const float *tau_given, std::vector<float> *retry_tau) {
  BatchGroupRun run;
  const int st = batch_group_queue(ix, c, run, queries, nq, limit, bf16, tau_given);
  if (st != VT_OK) {
    (void)hipStreamSynchronize(c.stream);  // (whatever was queued reads the context's pinned blocks)
    return st;
  }
  return batch_group_finish(ix, c, run, out, done, retry_tau);
}

// K1m serves a batch when every query's list fits its small wave buffers.
bool multi_scan_applies(const Shard *ix, size_t limit) {
  return limit >= 1 && std::min<size_t>(limit, ix->n) <= vt::scan_multi_max_k(vt::kMultiMaxQueries) &&
         !(ix->metric == VT_JACCARD && ix->dim >= 4096) && !vt::env::on(vt::env::NO_MULTI_SCAN);
}
double multi_scan_seconds(const Shard *ix, size_t nq) {
  const double sweeps = std::ceil((double)nq / vt::kMultiMaxQueries);
  const double waves = (double)ix->ctx.num_cus * 2 * vt::kWavesPerBlock;  // two blocks per CU
  const double tiles = std::ceil((double)ix->n / vt::scan_multi_tile_rows(vt::kMultiMaxQueries));
  double steps = std::ceil(tiles / waves) * std::ceil((double)ix->ld / 256.0);  // (tile, panel) steps of one wave
  double per_step = kMultiPanelS + 0.3e-6 / std::ceil((double)ix->ld / 256.0);
  if (ix->dim % 64 != 0) per_step *= 1.1;  // the variants that carry the tail handling
  return kMultiFixedS + sweeps * (kMultiSweepFixedS + steps * per_step + std::min(steps, 20.0) * kMultiRampS);
}

// `count` queries (rows `which[i]` of `queries`) in ceil(count / 8) sweeps of the corpus (K1m),
// every sweep and one batched select queued before the single wait.  Ranks strictly current.
int multi_scan_group(Shard *ix, Ctx &c, const float *queries, const std::vector<size_t> &which, size_t limit, vt_hits **out) {
  const uint32_t d = (uint32_t)ix->dim, ld = ix->ld, n = ix->n;
  const uint32_t k = (uint32_t)std::min<size_t>(limit, n);
  const size_t nq = which.size();
  const size_t lds = vt::scan_multi_lds_bytes(d, k, ix->metric);
  const uint32_t ntiles = (n + vt::scan_multi_tile_rows(vt::kMultiMaxQueries) - 1) / vt::scan_multi_tile_rows(vt::kMultiMaxQueries);
  const uint32_t blocks = c.grid_for(ntiles, lds, vt::scan_multi_blocks_per_cu(d, k, ix->metric));
  // (a sweep always reads a full group of query rows: the last group is padded with zero rows)
  const size_t nq_pad = (nq + vt::kMultiMaxQueries - 1) / vt::kMultiMaxQueries * vt::kMultiMaxQueries;
  VT_TRY(c.dBQ.ensure(nq_pad * ld));
  VT_TRY(c.hBQ.ensure(nq_pad * ld));
  VT_TRY(c.dPartKeys.ensure(nq * blocks * k));
  VT_TRY(c.dPartPay.ensure(nq * blocks * k));
  // per query a packed result block: 16-byte header + k entries (Entry is 16 bytes)
  const uint32_t out_stride = 16 + k * (uint32_t)sizeof(vt::Entry);
  VT_TRY(c.dBOut.ensure(nq * (k + 1)));
  VT_TRY(c.hBOut.ensure(nq * (k + 1)));
  std::memset(c.hBQ.p, 0, nq_pad * ld * sizeof(float));
  std::vector<uint32_t> qnz(nq, 0);
  for (size_t i = 0; i < nq; ++i) {
    const float *q = queries + which[i] * d;
    std::memcpy(c.hBQ.p + i * ld, q, (size_t)d * sizeof(float));
    for (uint32_t j = 0; j < d; ++j) qnz[i] += q[j] != 0.0f ? 1u : 0u;
  }
  VT_HIP(hipMemcpyAsync(c.dBQ.p, c.hBQ.p, nq_pad * ld * sizeof(float), hipMemcpyHostToDevice, c.stream));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
  uint32_t sweeps = 0;
  for (size_t g0 = 0; g0 < nq; g0 += vt::kMultiMaxQueries, ++sweeps) {
    const uint32_t gn = (uint32_t)std::min<size_t>(vt::kMultiMaxQueries, nq - g0);
    vt::MultiScanArgs a{};
    a.X = ix->dX;
    a.stride = ix->ld;
    a.Q = c.dBQ.p + g0 * ld;
    a.id_rank = ix->dRank.p;
    a.n = n;
    a.d = d;
    a.ld = ld;
    a.metric = ix->metric;
    a.order = ix->order;
    a.k = k;
    a.nq = gn;
    a.first_query = (uint32_t)g0;
    a.dbg = (uint32_t)vt::env::get(vt::env::MQ_DBG);
    for (uint32_t i = 0; i < gn; ++i) a.q_nonzero[i] = qnz[g0 + i];
    a.part_keys = c.dPartKeys.p;
    a.part_pay = c.dPartPay.p;
    a.status = c.dStatus.p;
    VT_HIP(vt::launch_scan_multi(a, blocks, c.stream));
  }
  if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
  VT_HIP(vt::launch_select_queries(c.dPartKeys.p, c.dPartPay.p, (uint32_t)nq, blocks * k, k, c.dBOut.p, out_stride, c.stream));
  int status = 0;
  VT_HIP(hipMemcpyAsync(c.hBOut.p, c.dBOut.p, nq * out_stride, hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipMemcpyAsync(&status, c.dStatus.p, sizeof(int), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipMemsetAsync(c.dStatus.p, 0, sizeof(int), c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  if (c.profiling) {
    float ms = 0.f;
    VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    c.prof.scan_launches += sweeps;
    c.prof.scan_ms += ms;
    c.prof.scan_rows += (uint64_t)sweeps * n;
    c.prof.scan_bytes += (uint64_t)sweeps * n * d * 4;
  }
  // "metric overflow" belongs to one query (flat.rs:105): the single-query path finds out whose
  if (status != 0) return kRetryInternal;
  for (size_t i = 0; i < nq; ++i) {
    const vt::Entry *blk = c.hBOut.p + i * (k + 1);  // [0] is the header
    uint32_t got = 0;
    std::memcpy(&got, reinterpret_cast<const unsigned char *>(blk) + 4, 4);
    got = std::min<uint32_t>(got, k);
    std::vector<vt::Entry> entries(blk + 1, blk + 1 + got);
    VT_TRY(make_hits(ix, entries, &out[which[i]]));
  }
  return VT_OK;
}

// Float hamming / jaccard batches on a current non-zero-bit column: up to eight queries per sweep of
// the column (K4p) when their lists fit its small wave buffers and the row length has an unrolled build.
bool pattern_group_applies(const Shard *ix, size_t limit) {
  const uint32_t words = ((uint32_t)ix->dim + 63) / 64;
  return pattern_search_applies(ix, limit) && !shard_stale(ix, NEED_NZBITS, limit) && limit >= 1 &&
         std::min<size_t>(limit, ix->n) <= (size_t)vt::kSmallK && vt::pattern_multi_supports((words + 1) / 2);
}

// `count` queries (rows `which[i]` of `queries`) in ceil(count / 8) sweeps of the non-zero-bit
// column, every sweep and one batched select queued before the single wait.  Ranks strictly current.
int pattern_scan_group(Shard *ix, Ctx &c, const float *queries, const std::vector<size_t> &which, size_t limit, vt_hits **out) {
  const uint32_t d = (uint32_t)ix->dim, n = ix->n;
  const uint32_t words = (d + 63) / 64, pairs = (words + 1) / 2;
  const uint32_t k = (uint32_t)std::min<size_t>(limit, n);
  const size_t nq = which.size();
  const size_t lds = vt::pattern_multi_lds_bytes();
  const uint32_t blocks = c.grid_for((n + 63) / 64, lds, 2);
  const size_t q_words = (size_t)2 * pairs;  // u64 per query
  const size_t up_floats = nq * q_words * 2;
  VT_TRY(c.dBQ.ensure(up_floats));
  VT_TRY(c.hBQ.ensure(up_floats));
  VT_TRY(c.dPartKeys.ensure(nq * blocks * k));
  VT_TRY(c.dPartPay.ensure(nq * blocks * k));
  const uint32_t out_stride = 16 + k * (uint32_t)sizeof(vt::Entry);  // per query a packed result block: header + k entries
  VT_TRY(c.dBOut.ensure(nq * (k + 1)));
  VT_TRY(c.hBOut.ensure(nq * (k + 1)));
  uint64_t *hbits = reinterpret_cast<uint64_t *>(c.hBQ.p);
  std::memset(hbits, 0, nq * q_words * sizeof(uint64_t));
  for (size_t i = 0; i < nq; ++i) {
    const float *q = queries + which[i] * d;
    uint64_t *w = hbits + i * q_words;
    for (uint32_t j = 0; j < d; ++j)
      if (q[j] != 0.0f) w[j / 64] |= 1ull << (j % 64);  // distances.rs:319-347: what the two metrics compare
  }
  VT_HIP(hipMemcpyAsync(c.dBQ.p, c.hBQ.p, nq * q_words * sizeof(uint64_t), hipMemcpyHostToDevice, c.stream));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
  uint32_t sweeps = 0;
  for (size_t g0 = 0; g0 < nq; g0 += vt::kPatternMultiMax, ++sweeps) {
    vt::PatternMultiArgs a{};
    a.bits = ix->dNzBits.p;
    a.qbits = reinterpret_cast<const uint64_t *>(c.dBQ.p) + g0 * q_words;
    a.id_rank = ix->dRank.p;
    a.n = n;
    a.words = words;
    a.pairs = pairs;
    a.d = d;
    a.k = k;
    a.nq = (uint32_t)std::min<size_t>(vt::kPatternMultiMax, nq - g0);
    a.first_query = (uint32_t)g0;
    a.jaccard = ix->metric == VT_JACCARD ? 1 : 0;
    a.part_keys = c.dPartKeys.p;
    a.part_pay = c.dPartPay.p;
    VT_HIP(vt::launch_pattern_multi(a, blocks, c.stream));
  }
  if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
  VT_HIP(vt::launch_select_queries(c.dPartKeys.p, c.dPartPay.p, (uint32_t)nq, blocks * k, k, c.dBOut.p, out_stride, c.stream));
  VT_HIP(hipMemcpyAsync(c.hBOut.p, c.dBOut.p, nq * out_stride, hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  if (c.profiling) {
    float ms = 0.f;
    VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    c.prof.hamming_launches += sweeps;
    c.prof.hamming_ms += ms;
    c.prof.hamming_bytes += (uint64_t)sweeps * n * words * 8;
    c.prof.hamming_queries += nq;
  }
  for (size_t i = 0; i < nq; ++i) {
    const vt::Entry *blk = c.hBOut.p + i * (k + 1);  // [0] is the header
    uint32_t got = 0;
    std::memcpy(&got, reinterpret_cast<const unsigned char *>(blk) + 4, 4);
    got = std::min<uint32_t>(got, k);
    std::vector<vt::Entry> entries(blk + 1, blk + 1 + got);
    VT_TRY(make_hits(ix, entries, &out[which[i]]));
  }
  return VT_OK;
}

// K2b applies wherever K2 does; which of the two nominates is the shard's setting
// (vt_flat_set_batch_nominate / VT_BATCH_NOMINATE, default bf16).
bool batch_nominates_bf16(const Shard *ix) { return ix->nominate == VT_NOMINATE_BF16; }

// True when a batch of nq queries takes the shared MFMA pass (and so needs the row norms).
bool batch_uses_mfma(const Shard *ix, size_t nq, size_t limit) {
  const bool mfma_metric = ix->metric == VT_COSINE || ix->metric == VT_INNER_PRODUCT ||
                           ix->metric == VT_NEG_INNER_PRODUCT || ix->metric == VT_L2 || ix->metric == VT_L2_SQUARED;
  // one shared pass over the corpus costs about 1.3 single scans (HBM-bound below 33 queries),
  // so it pays from two queries on
  bool use_mfma = mfma_metric && nq >= 2 && limit <= (size_t)vt::kMaxFusedK && limit > 0 && ix->n >= 4096 &&
                  !vt::env::on(vt::env::BATCH_NO_MFMA);
  if (use_mfma && !vt::env::on(vt::env::FORCE_BATCH_MFMA)) {  // (tests force the shared pass on small corpora)
    // nq single scans against one shared pass (K2: HBM-bound below ~33 queries, then MFMA-bound;
    // K2b: HBM-bound at every batch size)
    const double bytes = (double)ix->n * ix->ld * 4.0;
    const bool bf16 = batch_nominates_bf16(ix);
    double nq_pad = bf16 ? (double)vt::batch_bf16_pad((uint32_t)std::min<size_t>(nq, 256)) : 32;
    while (nq_pad < (double)std::min<size_t>(nq, 256)) nq_pad *= 2;
    const double groups = std::ceil((double)nq / 256.0);
    // (K2b: the pass streams at ~0.85 of a plain scan's rate -- the matrix pipe is busy 60 % of
    // the time beside it -- and every query adds a few hundred candidates to re-score: 6.5-6.9 ms
    // per 256 queries at 30 GB where K1m's sweep of eight takes 5.5)
    // (64 / 128 columns: 4.70 / 4.79 ms per pass at 30.72 GB against 4.48 for a plain scan; 256: 5.3-5.45)
    const double k2b_stream = nq_pad <= 64 ? 1.05 : nq_pad <= 128 ? 1.08 : 1.2;
    const double t_pass = bf16 ? std::max(k2b_stream * bytes / kScanBytesPerS, 2.0 * ix->n * nq_pad * ix->ld / kNominateFlopsPerS) +
                                     (double)std::min<size_t>(nq, 256) * 2.5e-6
                               : std::max(1.3 * bytes / kScanBytesPerS, 2.0 * ix->n * nq_pad * ix->ld / kBatchFlopsPerS);
    double t_other = (double)nq * scan_seconds(bytes);
    if (multi_scan_applies(ix, limit)) t_other = std::min(t_other, multi_scan_seconds(ix, nq));
    use_mfma = t_other > groups * (kBatchFixedS + t_pass);
  }
  return use_mfma;
}

bool funnel_group_applies(const Shard *ix, const size_t *stages, size_t nstages, size_t candidates, size_t limit);
int funnel_group(Shard *ix, Ctx &c, const float *queries, const std::vector<size_t> &which, const size_t *stages, size_t nstages,
                 size_t candidates, size_t limit, vt_hits **out, std::vector<char> &done, bool as_scan = false);
int funnel_groups(Shard *ix, Ctx &c, const float *queries, const std::vector<std::vector<size_t>> &groups, const size_t *stages,
                  size_t nstages, size_t candidates, size_t limit, vt_hits **out, std::vector<char> &done, bool as_scan);

// A batch of plain searches as K1p sweeps (funnel_group with the whole row as its one stage and `limit` candidates:
// the sweep's arithmetic over all d coordinates is K1's, the lists' `limit` best are flat_search's hits).  Measured
// beside K1m (tools/prefix_multi_probe.py, 8 queries per sweep, N x d x 4 = 7.7 GB): 4.3 / 5.1 / 5.8 / 5.9 TB/s at
// d = 64 / 128 / 320 / 640 where K1m walks 2.5 / 3.4 / 4.2 / 4.5 -- rows that are not whole 256-float panels -- and
// level with it where they are (d = 768: 5.21 against 5.24-5.37 ms); and it takes lists of up to 256, K1m's wave
// buffers 32.  So: rows off K1m's panel grid, or lists K1m cannot hold; from 64 MB of rows (six launches and a
// sample pass instead of two launches).  r05: on 32-float panels K1p is 7 % faster than it was at full width -- at
// d = 768, N = 10 M: 4.74 / 5.74 / 4.84 ms per sweep (dot / L2 / L1) where K1m takes 5.45 / 6.13 / 5.31 -- so wide rows of
// a corpus of 2 GB and more go to K1p as well (at d = 384 the two are level: K1m keeps those, with its two launches).
bool sweep_group_applies(const Shard *ix, size_t limit) {
  if (!vt::prefix_multi_supports(ix->metric)) return false;
  const size_t stage = (size_t)ix->dim;
  if (!funnel_group_applies(ix, &stage, 1, limit, limit)) return false;
  if (vt::env::on(vt::env::FORCE_SWEEP_GROUPS)) return true;  // (tests and soaks on corpora of a few MB)
  const double bytes = (double)ix->n * ix->ld * 4.0;
  if (bytes < 64e6) return false;
  return ix->ld % 256 != 0 || !multi_scan_applies(ix, limit) || (bytes >= 2e9 && ix->ld >= 512);
}

// a status that says "a buffer could not be had" (hipMalloc / hipHostMalloc refused, or the host's allocator did)
bool allocation_failed(int st, const std::string &why) {
  return st == VT_ERR_NOMEM || (st == VT_ERR_DEVICE && (why.find("hipMalloc") != std::string::npos || why.find("hipHostMalloc") != std::string::npos));
}

// Rank column strictly current, norms current when batch_uses_mfma (shard_prepare).
int batch_ready(Shard *ix, Ctx &c, const float *queries, size_t nq, size_t d, size_t limit, vt_hits **out) {
  // every query is validated like flat_search would (flat.rs:97-101), in order
  if (limit == 0) {
    for (size_t i = 0; i < nq; ++i) VT_TRY(empty_hits(&out[i]));
    return VT_OK;
  }
  for (size_t i = 0; i < nq; ++i) VT_TRY(validate_vector(queries + i * d, d, ix->dim));
  if (ix->n == 0) {
    for (size_t i = 0; i < nq; ++i) VT_TRY(empty_hits(&out[i]));
    return VT_OK;
  }
  std::vector<char> done(nq, 0);
  const bool use_mfma = batch_uses_mfma(ix, nq, limit);
  const bool bf16 = batch_nominates_bf16(ix);
  if (use_mfma) {
    std::vector<float> tau2(nq, std::numeric_limits<float>::quiet_NaN());
    std::vector<std::pair<size_t, size_t>> groups;  // (first query, queries)
    for (size_t g0 = 0; g0 < nq; g0 += 256) {
      const size_t gn = std::min<size_t>(256, nq - g0);
      if (gn >= 2) groups.emplace_back(g0, gn);  // a lone trailing query takes the single-query path below
    }
    // Several groups (config 3 as BASELINE.json writes it is 16 x 256 in one call): they alternate between this
    // context and a second one, and group g + 1 is staged and queued BEFORE group g is waited for -- its upload,
    // sample pass and threshold kernel run beside group g's exact rescoring and select, its pass over the rows
    // starts the moment the card is free, and group g's host side (the wait, 256 verdicts, 256 hit lists: 0.25 ms
    // of a 4.4-ms group) runs under it.  Each group's results are what it gives alone: nothing is shared but the card.
    SpareCtxLease spare(groups.size() >= 2 && !vt::env::on(vt::env::NO_GROUP_PIPELINE) ? ix : nullptr);
    Ctx *cx[2] = {&c, spare.c ? spare.c : &c};
    BatchGroupRun runs[2];
    auto settle = [&](size_t g, const std::vector<char> &gdone, const std::vector<float> &gtau) {
      for (size_t i = 0; i < groups[g].second; ++i) done[groups[g].first + i] = gdone[i];
      for (size_t i = 0; i < gtau.size(); ++i) tau2[groups[g].first + i] = gtau[i];
      // (a shard of a sharded handle: the lists this group settled are final -- nothing below rewrites a done query --
      // and the handle's calling thread may start merging them with the other shards' while the next groups run)
      if (const MergeRequest *mr = merge_request_of(ix))
        for (size_t i = 0; i < groups[g].second; ++i)
          if (gdone[i]) mr->final_flags[groups[g].first + i].store(1, std::memory_order_release);
    };
    // groups [in_series, end) run one after the other on `c`: all of them when there is no second context, and whatever is
    // left when the second context could not get its scratch (below)
    size_t in_series = spare.c ? groups.size() : 0;
    if (spare.c) {
      VT_TRY(reader_ready(ix, *spare.c));
      // (kBatchTailCus: CUs every pass but the last leaves to the groups around it; the four-stage ring between groups:
      // both measured in DESIGN 5.1 / A.15)
      auto queue = [&](size_t g) {
#ifdef VT_TEST_HOOKS
        if ((g & 1) == 1 && vt::env::on(vt::env::TEST_REFUSE_SPARE_SCRATCH))
          return fail(VT_ERR_DEVICE, "hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T)): out of memory (injected)");
#endif
        return batch_group_queue(ix, *cx[g & 1], runs[g & 1], queries + groups[g].first * d, groups[g].second, limit, bf16, nullptr,
                                 kBatchTailCus, /*four_stages=*/true);
      };
      int st = queue(0);
      for (size_t g = 0; g < groups.size() && st == VT_OK; ++g) {
        const int st_next = g + 1 < groups.size() ? queue(g + 1) : VT_OK;
        const std::string why_next = st_next != VT_OK ? g_last_error : std::string();
        std::vector<char> gdone(groups[g].second, 0);
        std::vector<float> gtau;
        st = batch_group_finish(ix, *cx[g & 1], runs[g & 1], out + groups[g].first, gdone, bf16 ? &gtau : nullptr);
        if (st == VT_OK) settle(g, gdone, gtau);
        // The second context roughly doubles a call's device and pinned scratch.  On a card nearly filled by corpus and
        // shadow its buffers may not fit where the one-context path of r04 did: then the spare is dropped and the rest of
        // the call runs in series on `c`, as it would have without one.  Nothing of the refused group is in
        // flight: every buffer is ensured before the first launch of a group.
        if (st == VT_OK && st_next != VT_OK && ((g + 1) & 1) == 1 && allocation_failed(st_next, why_next)) {
          (void)hipStreamSynchronize(cx[1]->stream);
          (void)hipGetLastError();
          in_series = g + 1;
          break;
        }
        if (st == VT_OK && st_next != VT_OK) g_last_error = why_next;
        if (st == VT_OK) st = st_next;
      }
      if (st != VT_OK) {  // (whatever is still queued reads and writes the two contexts' pinned blocks)
        const std::string why = g_last_error;
        (void)hipStreamSynchronize(cx[0]->stream);
        (void)hipStreamSynchronize(cx[1]->stream);
        (void)hipGetLastError();
        g_last_error = why;
        return st;
      }
    }
    for (size_t g = in_series; g < groups.size(); ++g) {
      std::vector<char> gdone(groups[g].second, 0);
      std::vector<float> gtau;
      VT_TRY(batch_group(ix, c, queries + groups[g].first * d, groups[g].second, limit, out + groups[g].first, gdone, bf16, nullptr,
                         bf16 ? &gtau : nullptr));
      settle(g, gdone, gtau);
    }
    // K2b's second pass: a query whose k exact hits did not clear tau by the margin names the
    // threshold that its k-th hit does clear; one more pass over the rows with those thresholds
    // certifies all such queries at once (K1m would need a sweep per eight of them).
    std::vector<size_t> again;
    for (size_t i = 0; i < nq; ++i)
      if (!done[i] && !std::isnan(tau2[i])) again.push_back(i);
    if (bf16 && again.size() >= 2) {
      for (size_t g0 = 0; g0 < again.size(); g0 += 256) {
        const size_t gn = std::min<size_t>(256, again.size() - g0);
        std::vector<float> qs(gn * d), taus(gn);
        std::vector<vt_hits *> outs(gn, nullptr);
        for (size_t i = 0; i < gn; ++i) {
          std::memcpy(qs.data() + i * d, queries + again[g0 + i] * d, d * sizeof(float));
          taus[i] = tau2[again[g0 + i]];
        }
        std::vector<char> gdone(gn, 0);
        VT_TRY(batch_group(ix, c, qs.data(), gn, limit, outs.data(), gdone, true, taus.data(), nullptr));
        for (size_t i = 0; i < gn; ++i)
          if (gdone[i]) {
            out[again[g0 + i]] = outs[i];
            done[again[g0 + i]] = 1;
          }
      }
    }
  }
  std::vector<size_t> left;
  for (size_t i = 0; i < nq; ++i)
    if (!done[i]) left.push_back(i);
  c.prof.batch_fallbacks += use_mfma ? left.size() : 0;
  // what the matrix cores did not take (no GEMM form for this metric, a small batch, a query
  // the bound could not certify): several queries per sweep of the corpus when their lists
  // fit, else one scan each
  // Float hamming / jaccard with a current non-zero-bit column: eight queries per sweep of the column
  // (K4p) when their lists fit ...
  if (left.size() >= 2 && pattern_group_applies(ix, limit)) {
    // (256 queries per call: the partial lists take blocks * k entries per query)
    for (size_t g0 = 0; g0 < left.size(); g0 += 256) {
      const std::vector<size_t> part(left.begin() + g0, left.begin() + std::min(left.size(), g0 + 256));
      VT_TRY(pattern_scan_group(ix, c, queries, part, limit, out));
    }
    return VT_OK;
  }
  // ... else each query is a K4 pass over 1/32 of the bytes a sweep of the rows reads (search_ready
  // below takes it) -- unless a sweep of the rows for eight queries is cheaper than eight such
  // passes with their ~50 us of launches, select and wait each, as it is on corpora below a GB or two.
  const double pattern_s = 50e-6 + (double)ix->n * (double)(((size_t)ix->dim + 63) / 64 * 8) / 5.5e12;
  const bool by_pattern = pattern_search_applies(ix, limit) && !shard_stale(ix, NEED_NZBITS, limit) &&
                          (left.size() < 2 || !multi_scan_applies(ix, limit) ||
                           (double)left.size() * pattern_s < multi_scan_seconds(ix, left.size()));
  if (!by_pattern && left.size() >= 2 && sweep_group_applies(ix, limit)) {
    const size_t stage = (size_t)ix->dim;
    std::vector<std::vector<size_t>> sweeps;
    for (size_t g0 = 0; g0 + 2 <= left.size(); g0 += vt::kPrefixMultiMax)
      sweeps.emplace_back(left.begin() + g0, left.begin() + std::min(left.size(), g0 + vt::kPrefixMultiMax));
    // (a group with an overflow somewhere comes back undone: its queries go on below, each reporting its own; the
    // groups alternate between two contexts, funnel_groups)
    VT_TRY(funnel_groups(ix, c, queries, sweeps, &stage, 1, limit, limit, out, done, true));
    std::vector<size_t> rest;
    for (size_t i : left)
      if (!done[i]) rest.push_back(i);
    left.swap(rest);
  }
  if (!by_pattern && left.size() >= 2 && multi_scan_applies(ix, limit) &&
      (multi_scan_seconds(ix, left.size()) < (double)left.size() * scan_seconds((double)ix->n * ix->ld * 4.0) ||
       vt::env::on(vt::env::FORCE_MULTI_SCAN))) {  // (tests force the sweep on corpora of a few thousand rows)
    const int st = multi_scan_group(ix, c, queries, left, limit, out);
    if (st == VT_OK) return VT_OK;
    if (st != kRetryInternal) return st;
    for (size_t i : left) {  // an overflow somewhere: one by one, so that it is reported for its own query's position
      delete out[i];
      out[i] = nullptr;
    }
  }
  for (size_t i : left) VT_TRY(search_ready(ix, c, queries + i * d, d, limit, &out[i]));
  return VT_OK;
}

}  // namespace
