// vt_batch_staged.h -- query batches (K2 / K1m host side) and the staged searches (quantized, funnel, hybrid) on one shard.
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
  const long tiles_setting = vt::env::get(vt::env::BATCH_SAMPLE_TILES);
  // (K2s: 512 tiles by default -- 33 us -> 60 us of sample pass, half the candidates to rescore behind the pass)
  SamplePlan sp = plan_sample(!shadow ? dense_cap : tiles_setting >= 64 && tiles_setting <= 512 ? (uint32_t)tiles_setting : 512u);
  if (!sp.by_maxima && sp.rows > 65536) sp = plan_sample(dense_cap);
  const uint32_t stride = sp.stride, ntiles_sample = sp.ntiles, sample_rows = sp.rows, rank = sp.rank, sample_groups = sp.groups;
  const bool by_maxima = sp.by_maxima;
  const uint32_t cand_cap = 8192;
  const uint32_t kBlocksPerQuery = [] {  // blocks of the exact rescoring per query (VT_RESCORE_BLOCKS: A/B)
    const long v = vt::env::get(vt::env::RESCORE_BLOCKS);
    return v >= 1 && v <= 64 ? (uint32_t)v : 8u;
  }();

  VT_TRY(c.dBQ.ensure((size_t)nq_pad * ld));
  VT_TRY(c.hBQ.ensure((size_t)nq_pad * ld));
  VT_TRY(c.dBTau.ensure(nq_pad));
  VT_TRY(c.hBTau.ensure(nq_pad));
  if (!tau_given) VT_TRY(c.dBSample.ensure((size_t)nq_pad * (by_maxima ? sample_groups : sample_rows)));
  VT_TRY(c.dBCand.ensure((size_t)nq_pad * cand_cap));
  VT_TRY(c.dBCount.ensure(nq_pad));
  VT_TRY(c.hBCount.ensure(nq_pad));
  VT_TRY(c.dBOut.ensure((size_t)nq_pad * k));
  VT_TRY(c.hBOut.ensure((size_t)nq_pad * k));
  VT_TRY(c.dBOutCount.ensure(nq_pad));
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
  (void)ld;
  (void)n;
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
         std::min<size_t>(limit, ix->n) <= (size_t)vt::kSmallK && vt::pattern_multi_supports((words + 1) / 2) &&
         !vt::env::on(vt::env::NO_PATTERN_GROUPS);
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

// A batch of plain searches as K1p sweeps (funnel_group with the whole row as its one stage and `limit` candidates:
// the sweep's arithmetic over all d coordinates is K1's, the lists' `limit` best are flat_search's hits).  Measured
// beside K1m (tools/prefix_multi_probe.py, 8 queries per sweep, N x d x 4 = 7.7 GB): 4.3 / 5.1 / 5.8 / 5.9 TB/s at
// d = 64 / 128 / 320 / 640 where K1m walks 2.5 / 3.4 / 4.2 / 4.5 -- rows that are not whole 256-float panels -- and
// level with it where they are (d = 768: 5.21 against 5.24-5.37 ms); and it takes lists of up to 256, K1m's wave
// buffers 32.  So: rows off K1m's panel grid, or lists K1m cannot hold; from 64 MB of rows (six launches and a
// sample pass instead of two launches).
bool sweep_group_applies(const Shard *ix, size_t limit) {
  if (!vt::prefix_multi_supports(ix->metric) || vt::env::on(vt::env::NO_SWEEP_GROUPS)) return false;
  const size_t stage = (size_t)ix->dim;
  if (!funnel_group_applies(ix, &stage, 1, limit, limit)) return false;
  if (vt::env::on(vt::env::FORCE_SWEEP_GROUPS)) return true;  // (tests and soaks on corpora of a few MB)
  if ((double)ix->n * ix->ld * 4.0 < 64e6) return false;
  return ix->ld % 256 != 0 || !multi_scan_applies(ix, limit);
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
    };
    if (!spare.c) {
      for (size_t g = 0; g < groups.size(); ++g) {
        std::vector<char> gdone(groups[g].second, 0);
        std::vector<float> gtau;
        VT_TRY(batch_group(ix, c, queries + groups[g].first * d, groups[g].second, limit, out + groups[g].first, gdone, bf16, nullptr,
                           bf16 ? &gtau : nullptr));
        settle(g, gdone, gtau);
      }
    } else {
      // (VT_BATCH_TAIL_CUS: CUs every pass but the last leaves to the groups around it; measured in DESIGN 5.1)
      const long tail_cus = vt::env::get(vt::env::BATCH_TAIL_CUS);
      const uint32_t idle = tail_cus >= 0 ? (uint32_t)tail_cus : kBatchTailCus;
      auto queue = [&](size_t g) {
        return batch_group_queue(ix, *cx[g & 1], runs[g & 1], queries + groups[g].first * d, groups[g].second, limit, bf16, nullptr,
                                 idle, !vt::env::on(vt::env::BATCH_PASS_FIVE));  // (VT_BATCH_PASS_FIVE=1: A/B)
      };
      int st = queue(0);
      for (size_t g = 0; g < groups.size() && st == VT_OK; ++g) {
        const int st_next = g + 1 < groups.size() ? queue(g + 1) : VT_OK;
        std::vector<char> gdone(groups[g].second, 0);
        std::vector<float> gtau;
        st = batch_group_finish(ix, *cx[g & 1], runs[g & 1], out + groups[g].first, gdone, bf16 ? &gtau : nullptr);
        if (st == VT_OK) settle(g, gdone, gtau);
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
    for (size_t g0 = 0; g0 + 2 <= left.size(); g0 += vt::kPrefixMultiMax) {
      const std::vector<size_t> which(left.begin() + g0, left.begin() + std::min(left.size(), g0 + vt::kPrefixMultiMax));
      const int st = funnel_group(ix, c, queries, which, &stage, 1, limit, limit, out, done, true);
      if (st == kRetryInternal) {  // an overflow somewhere: these go on below, each reporting its own
        for (size_t i : which) {
          delete out[i];
          out[i] = nullptr;
          done[i] = 0;
        }
      } else if (st != VT_OK) {
        return st;
      }
    }
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
  VT_TRY(upload_query(c, query, n, &qnz, true, /*direct=*/candidates <= (size_t)vt::kSelListMax));
  const size_t ncand = std::min<size_t>(candidates, ix->n);
  const size_t keep = local ? ncand : limit;
  // K4h needs integer bins in LDS, one fused select, and enough rows to be worth two passes
  const bool hist_ok = ncand <= (size_t)vt::kSelListMax && d <= vt::kHammingHistMaxDim &&
                       (ix->n >= 16384 || ncand > (size_t)vt::kMaxFusedK) &&
                       !vt::env::on(vt::env::HAMMING_LISTS);
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
         vt::hamming_multi_lds_bytes(d, (d + 63) / 64, 2) <= 64 * 1024 && !vt::env::on(vt::env::HAMMING_LISTS) &&
         !vt::env::on(vt::env::NO_QUANTIZED_GROUPS) &&
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
      const uint32_t nslots = (uint32_t)groups.size();
      const uint32_t k2 = (uint32_t)std::min<size_t>(limit, std::min<size_t>(candidates, ix->n));
      std::vector<int> queued(groups.size(), VT_OK);
      for (uint32_t g = 0; g < nslots; ++g) {
        queued[g] = quantized_group(ix, c, queries, groups[g], candidates, limit, out, g, nslots, true);
        if (queued[g] != VT_OK && queued[g] != kRetryInternal) {
          (void)hipStreamSynchronize(c.stream);
          return queued[g];
        }
      }
      VT_HIP(hipStreamSynchronize(c.stream));
      for (uint32_t g = 0; g < nslots; ++g) {
        const int st = queued[g] == VT_OK ? quantized_group_finish(ix, c, quantized_group_slot(ix, g, nslots), groups[g], k2, out)
                                          : queued[g];
        VT_TRY(settle(groups[g], st));
      }
    } else {
      for (const auto &which : groups) {
        const auto tg = std::chrono::steady_clock::now();
        const int st = quantized_group(ix, c, queries, which, candidates, limit, out);
        if (vt::env::on(vt::env::TRACE_QGROUP))
          std::fprintf(stderr, "[vt] quantized group of %zu: status %d, %.3f ms\n", which.size(), st,
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tg).count());
        VT_TRY(settle(which, st));
      }
    }
  }
  for (size_t i = 0; i < nq; ++i)
    if (!done[i]) VT_TRY(quantized_ready(ix, c, queries + i * d, d, candidates, limit, &out[i]));
  return VT_OK;
}

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
  VT_TRY(upload_query(c, query, n, &qnz_full, pattern_metric(ix->metric) ? 2 : 0, /*direct=*/stages[0] <= 256 && candidates <= (size_t)vt::kSelListMax));
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
  if (nstages == 0 || vt::env::on(vt::env::NO_FUNNEL_GROUPS)) return false;
  if (ix->metric != VT_COSINE && !vt::prefix_multi_supports(ix->metric)) return false;
  const size_t k1 = std::min<size_t>(candidates, ix->n);
  if (ix->n < 16384 || k1 == 0 || k1 > (size_t)vt::kMaxFusedK || limit == 0) return false;
  if (ix->metric != VT_COSINE) return vt::scan_lds_bytes((uint32_t)ix->dim, (uint32_t)k1) != 0;
  return (size_t)2 * (((size_t)ix->dim + 3) / 4 * 4) * 4 <= 160 * 1024;  // the rerank keeps row + query in LDS
}

int funnel_group(Shard *ix, Ctx &c, const float *queries, const std::vector<size_t> &which, const size_t *stages,
                 size_t nstages, size_t candidates, size_t limit, vt_hits **out, std::vector<char> &done, bool as_scan) {
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
  a.sample_rows = pa.sample_rows = sample_rows;
  if (cosine) VT_HIP(vt::launch_cosine_scan_multi(a, c.grid_for(stiles, lds), c.stream));
  else VT_HIP(vt::launch_prefix_multi(pa, c.grid_for(stiles, lds), c.stream));
  VT_HIP(vt::launch_sample_tau(c.dBSample.p, sample_rows, vt::kCosineMultiMax, nq, rank, c.dBTau.p, c.stream));
  // pass 1: every row's prefix once; (query, row) pairs at or above the thresholds into the lists
  a.sample = pa.sample = nullptr;
  a.tau = pa.tau = c.dBTau.p;
  a.cand_keys = pa.cand_keys = c.dPartKeys.p;
  a.cand_pay = pa.cand_pay = c.dPartPay.p;
  a.cand_count = pa.cand_count = c.dBCount.p;
  a.cand_cap = pa.cand_cap = kListCap;
  if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
  if (cosine) VT_HIP(vt::launch_cosine_scan_multi(a, c.grid_for(ntiles, lds), c.stream));
  else VT_HIP(vt::launch_prefix_multi(pa, c.grid_for(ntiles, lds), c.stream));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
  VT_HIP(hipMemcpyAsync(hListCount, c.dBCount.p, vt::kCosineMultiMax * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(vt::launch_select_lists(c.dPartKeys.p, c.dPartPay.p, nq, kListCap, c.dBCount.p, k1, c.dStageB.p,
                                 (uint32_t)sizeof(ResultBlock), c.stream));
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
    for (size_t g0 = 0; g0 < nq; g0 += vt::kCosineMultiMax) {
      std::vector<size_t> which;
      for (size_t i = g0; i < std::min<size_t>(nq, g0 + vt::kCosineMultiMax); ++i) which.push_back(i);
      if (which.size() < 2) break;
      const int st = funnel_group(ix, c, queries, which, stages, nstages, candidates, limit, out, done);
      if (st == kRetryInternal) {
        for (size_t i : which) {
          delete out[i];
          out[i] = nullptr;
          done[i] = 0;
        }
      } else if (st != VT_OK) {
        return st;
      }
    }
  }
  for (size_t i = 0; i < nq; ++i)
    if (!done[i]) VT_TRY(funnel_ready(ix, c, queries + i * d, d, stages, nstages, candidates, limit, &out[i]));
  return VT_OK;
}

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
