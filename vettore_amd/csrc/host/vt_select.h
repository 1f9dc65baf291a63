// vt_select.h -- launching the scans and cutting their lists: select passes, the radix threshold, Hamming and query upload.
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// Internal (never crosses the ABI): a device-side list overflowed, redo on the general path.
constexpr int kRetryInternal = -100;

// Rough per-call times on MI355X (tools/size_probe.py, tools/latency_floor.py), used only
// to choose between equivalent code paths: one fused scan of `bytes`, and the fixed cost
// the multi-kernel paths add on top of their scans.
constexpr double kScanFixedS = 35e-6, kScanBytesPerS = 6.5e12;
constexpr double kThresholdFixedS = 140e-6, kBatchFixedS = 180e-6, kBatchFlopsPerS = 135e12;
constexpr double kNominateFlopsPerS = 1.0e15;  // K2b's bf16 pass when it is not HBM-bound (it is, at every shape measured)
// K1m: one sweep carries up to 8 queries; a chain of sweeps pays the call's fixed cost once.  A
// sweep is priced per (tile, 256-float panel) a resident wave works through -- 2.9 us each once
// the chip streams, 4.5 us for a wave's first ones -- plus its prologue and list merges
// (tools/batch_path_probe.py, tools/multi_probe.py: 68 us at 150 MB, 326 us at 1.5 GB, 5.46 ms at 30 GB of d=768 rows)
constexpr double kMultiFixedS = 50e-6, kMultiSweepFixedS = 45e-6, kMultiPanelS = 2.9e-6, kMultiRampS = 1.6e-6;
inline double scan_seconds(double bytes) { return kScanFixedS + bytes / kScanBytesPerS; }

// ------------------------------------------------------------------ selection
// One select launch + stream sync; the k winners arrive in c.hRes (pinned,
// written by the kernel through the host mapping).
int select_pass(Ctx &c, const uint64_t *keys, const vt::Payload *pay, uint32_t m, uint32_t k, uint64_t lo, bool has_lo) {
  VT_HIP(vt::launch_select(keys, pay, m, k, lo, has_lo ? 1 : 0, c.dStatus.p, c.dResMapped, c.dSelKeys.p, c.dSelPay.p, c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  return VT_OK;
}

// Selects the `want` smallest keys among `m` candidates already on the device
// (rerank candidates) and appends them to `out` in ascending order.
int collect_from_keys(Ctx &c, const uint64_t *keys, const vt::Payload *pay, uint32_t m, size_t want,
                      std::vector<vt::Entry> &out) {
  const size_t goal = out.size() + std::min<size_t>(want, m);
  uint64_t lo = 0;
  bool has_lo = false;
  while (out.size() < goal) {
    const uint32_t k = (uint32_t)std::min<size_t>((size_t)vt::kMaxFusedK, goal - out.size());
    VT_TRY(select_pass(c, keys, pay, m, k, lo, has_lo));
    if (c.hRes.p->status == vt::kStatusRetry) return kRetryInternal;
    if (c.hRes.p->status == VT_ERR_OVERFLOW) return VT_ERR_OVERFLOW;
    const uint32_t got = c.hRes.p->count;
    for (uint32_t i = 0; i < got; ++i) out.push_back(c.hRes.p->e[i]);
    if (got < k) break;
    lo = c.hRes.p->e[got - 1].key;
    has_lo = true;
  }
  return VT_OK;
}

// All `m` <= kSelListMax candidates (keys/payload on the device) in ascending key order,
// appended to `out`: one launch, one wait.
int collect_sorted_list(Ctx &c, const uint64_t *keys, const vt::Payload *pay, uint32_t m, std::vector<vt::Entry> &out) {
  const size_t bytes = 16 + (size_t)vt::kSelListMax * sizeof(vt::Entry);
  if (!c.dBigMapped) {
    VT_TRY(c.hBig.ensure(bytes));
    VT_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&c.dBigMapped), c.hBig.p, 0));
  }
  auto *head = reinterpret_cast<vt::BigResultHeader *>(c.dBigMapped);
  auto *ents = reinterpret_cast<vt::Entry *>(c.dBigMapped + 16);
  VT_HIP(vt::launch_sort_list(keys, pay, m, c.dStatus.p, head, ents, c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  const auto *hh = reinterpret_cast<const vt::BigResultHeader *>(c.hBig.p);
  if (hh->status == vt::kStatusRetry) return kRetryInternal;
  if (hh->status == VT_ERR_OVERFLOW) return VT_ERR_OVERFLOW;
  const auto *he = reinterpret_cast<const vt::Entry *>(c.hBig.p + 16);
  out.insert(out.end(), he, he + hh->count);
  return VT_OK;
}

struct ScanJob {
  const float *X;
  size_t stride;
  const uint32_t *id_rank;
  const uint32_t *gather;
  uint32_t gather_stride;
  uint32_t n;
  uint32_t d;
  int metric;
  int order;
  uint32_t q_nonzero;
};

// Limits above kMaxFusedK in ONE scan: the scan has written a key per row into c.dKeyCol;
// three radix passes + a collect pass + one list select leave the exact `k` best rows,
// unsorted, in c.dListPay (Payload.row = position in the key column).
constexpr size_t kThresholdMinRows = 16384;
constexpr uint32_t kThresholdListCap = 65536;

// One scan + radix threshold, or one scan per 256 hits?  Whichever the model says is shorter.
bool threshold_applies(size_t total, uint32_t n, double scan_bytes, double pass_fixed_s = kScanFixedS) {
  if (total <= (size_t)vt::kMaxFusedK || total > (size_t)vt::kSelListMax || n < kThresholdMinRows)
    return false;
  if (vt::env::on(vt::env::FORCE_THRESHOLD_SELECT)) return true;  // tests: exercise the path on small corpora
  const double passes = std::ceil((double)total / vt::kMaxFusedK);
  const double t_pass = pass_fixed_s + scan_bytes / kScanBytesPerS;
  const double t_loop = passes * t_pass;
  const double t_threshold = t_pass + kThresholdFixedS + passes * 25e-6;
  return t_threshold < t_loop;
}

// Limits above kSelListMax (flat.ex:98-103 allows any limit below 2^32) in ONE scan: key and
// payload columns, an EXACT threshold on the device -- six radix passes resolve all 64 bits of
// the k-th smallest key, so with the unique keys of a strictly ranked index the collect pass
// leaves exactly the k winners however many rows tie in their rank (three passes, r02's rule,
// left every key sharing the k-th one's 33-bit prefix: float hamming at limit 1000 overflowed the
// list and fell to one scan per 256 hits) --, the list handed to the host as it is and ordered
// there on several threads.  limit >= rows: there is nothing to cut, the two columns leave the
// device whole.  kRetryInternal: more entries than asked for (equal keys: unranked rows) -- the
// caller takes the pass-per-256 loop.
int threshold_big(Ctx &c, const vt::ScanArgs &scan, uint32_t blocks, uint32_t n, uint32_t k, bool timed, uint32_t d,
                  std::vector<vt::Entry> &out) {
  const bool all_rows = k >= n;
  const size_t cap = all_rows ? (size_t)n : (size_t)k + 4096;  // (slack: equal keys are reported, not silently cut)
  VT_TRY(c.dKeyCol.ensure(((size_t)n + 1) / 2 * 2));
  VT_TRY(c.dPayCol.ensure(n));
  VT_TRY(c.dRadixHist.ensure(6 * vt::kRadixBins));
  VT_TRY(c.dRadixCount.ensure(1));
  if (!all_rows) {
    VT_TRY(c.dPartKeys.ensure(cap));
    VT_TRY(c.dPartPay.ensure(cap));
  }
  VT_TRY(c.hListKeys.ensure(cap));
  VT_TRY(c.hListPay.ensure(cap));
  vt::ScanArgs a = scan;
  a.k = 1;
  a.key_out = c.dKeyCol.p;
  a.pay_out = c.dPayCol.p;
  if (timed) VT_HIP(hipEventRecord(c.ev0, c.stream));
  VT_HIP(vt::launch_scan(a, blocks, c.stream));
  if (timed) VT_HIP(hipEventRecord(c.ev1, c.stream));
  uint32_t count = 0;
  int status = 0;
  if (!all_rows) {
    VT_HIP(hipMemsetAsync(c.dRadixHist.p, 0, 6 * vt::kRadixBins * sizeof(uint32_t), c.stream));
    vt::RadixArgs r{};
    r.keys = c.dKeyCol.p;
    r.n = n;
    r.k = k;
    r.hist = c.dRadixHist.p;
    r.list_count = c.dRadixCount.p;
    r.list_keys = c.dPartKeys.p;
    r.list_pay = c.dPartPay.p;
    r.cap = (uint32_t)cap;
    r.status = c.dStatus.p;
    r.pay_col = c.dPayCol.p;
    r.passes = 6;
    const uint32_t rblocks = (uint32_t)c.num_cus * 8;
    for (int pass = 0; pass < 6; ++pass) VT_HIP(vt::launch_radix_pass(r, pass, rblocks, c.stream));
    VT_HIP(vt::launch_radix_collect(r, rblocks, c.stream));
    VT_HIP(hipMemcpyAsync(&count, c.dRadixCount.p, sizeof(count), hipMemcpyDeviceToHost, c.stream));
  }
  VT_HIP(hipMemcpyAsync(&status, c.dStatus.p, sizeof(status), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipMemsetAsync(c.dStatus.p, 0, sizeof(int), c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  if (timed) {
    float ms = 0.f;
    VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    c.prof.scan_launches += 1;
    c.prof.scan_ms += ms;
    c.prof.scan_rows += n;
    c.prof.scan_bytes += (uint64_t)n * d * 4;
  }
  if (status == VT_ERR_OVERFLOW) return VT_ERR_OVERFLOW;
  if (all_rows) count = n;
  else if (status == vt::kStatusRetry || count > cap) return kRetryInternal;
  const uint64_t *src_keys = all_rows ? c.dKeyCol.p : c.dPartKeys.p;
  const vt::Payload *src_pay = all_rows ? c.dPayCol.p : c.dPartPay.p;
  VT_HIP(hipMemcpyAsync(c.hListKeys.p, src_keys, (size_t)count * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipMemcpyAsync(c.hListPay.p, src_pay, (size_t)count * sizeof(vt::Payload), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  std::vector<vt::Entry> list;
  list.reserve(count);
  for (uint32_t i = 0; i < count; ++i) {
    if (c.hListKeys.p[i] == vt::kEmptyKey) continue;
    vt::Entry e;
    e.key = c.hListKeys.p[i];
    e.row = c.hListPay.p[i].row;
    e.raw = c.hListPay.p[i].raw;
    list.push_back(e);
  }
  const size_t take = std::min<size_t>(k, list.size());
  parallel_sort(list, [](const vt::Entry &x, const vt::Entry &y) { return x.key < y.key; });
  list.resize(take);
  out = std::move(list);
  return VT_OK;
}

int threshold_rows(Ctx &c, uint32_t n, uint32_t k) {
  VT_TRY(c.dRadixHist.ensure(6 * vt::kRadixBins));
  VT_TRY(c.dRadixCount.ensure(1));
  VT_TRY(c.dPartKeys.ensure(kThresholdListCap));
  VT_TRY(c.dPartPay.ensure(kThresholdListCap));
  VT_TRY(c.dListKeys.ensure(k));
  VT_TRY(c.dListPay.ensure(k));
  VT_HIP(hipMemsetAsync(c.dRadixHist.p, 0, 6 * vt::kRadixBins * sizeof(uint32_t), c.stream));
  vt::RadixArgs r{};
  r.keys = c.dKeyCol.p;
  r.n = n;
  r.k = k;
  r.hist = c.dRadixHist.p;
  r.list_count = c.dRadixCount.p;
  r.list_keys = c.dPartKeys.p;
  r.list_pay = c.dPartPay.p;
  r.cap = kThresholdListCap;
  r.status = c.dStatus.p;
  // all 64 bits: exactly the k smallest keys reach the list, whatever the number of rows that tie
  // in their rank (float hamming / jaccard take a handful of distinct values: with the 33-bit
  // threshold of r02 a limit of 1000 collected every row of the k-th value, overflowed the list and
  // cost one scan per 256 hits -- 1.15 ms where the other metrics took 0.28)
  r.passes = 6;
  const uint32_t blocks = (uint32_t)c.num_cus * 8;
  for (int pass = 0; pass < 6; ++pass) VT_HIP(vt::launch_radix_pass(r, pass, blocks, c.stream));
  VT_HIP(vt::launch_radix_collect(r, blocks, c.stream));
  VT_HIP(vt::launch_select_list(c.dPartKeys.p, c.dPartPay.p, kThresholdListCap, c.dRadixCount.p, k, c.dListKeys.p,
                                c.dListPay.p, c.stream));
  return VT_OK;
}

// Scan + select passes until `want` hits are collected (ascending by key).
// The query must already be in c.dQ (padded to padded_dim(d)).
int run_scan(Ctx &c, const ScanJob &j, size_t want, std::vector<vt::Entry> &out, bool count_profile) {
  if (vt::scan_lds_bytes(j.d, 1) == 0)
    return fail(VT_ERR_UNSUPPORTED, "dimension " + std::to_string(j.d) + " exceeds what the scan kernel stages in LDS");
  const uint32_t tile_rows = vt::scan_tile_rows(j.n, j.d, c.resident_waves());
  const uint32_t ntiles = (j.n + tile_rows - 1) / tile_rows;
  // very wide rows leave no LDS for the large candidate buffer: smaller passes
  const size_t kmax = vt::scan_lds_bytes(j.d, vt::kMaxFusedK) ? (size_t)vt::kMaxFusedK : (size_t)vt::kSmallK;
  uint64_t lo = 0;
  bool has_lo = false;
  const size_t total = std::min<size_t>(want, j.n);
  if (!j.gather && out.empty() && total > (size_t)vt::kSelListMax && j.n >= kThresholdMinRows) {
    vt::ScanArgs a{};
    a.X = j.X;
    a.stride = j.stride;
    a.q = c.qsrc;
    a.id_rank = j.id_rank;
    a.n = j.n;
    a.d = j.d;
    a.metric = j.metric;
    a.order = j.order;
    a.q_nonzero = j.q_nonzero;
    a.tile_rows = tile_rows;
    a.part_keys = c.dPartKeys.p;  // (unused in key-column mode)
    a.part_pay = c.dPartPay.p;
    a.status = c.dStatus.p;
    VT_TRY(c.dPartKeys.ensure(kThresholdListCap));
    VT_TRY(c.dPartPay.ensure(kThresholdListCap));
    a.part_keys = c.dPartKeys.p;
    a.part_pay = c.dPartPay.p;
    const int rc = threshold_big(c, a, c.grid_for(ntiles, vt::scan_lds_bytes(j.d, 1)), j.n, (uint32_t)total,
                                 c.profiling && count_profile, j.d, out);
    if (rc != kRetryInternal) return rc;
    out.clear();  // more equal keys at the threshold than the list holds: the pass-per-256 loop below
  }
  if (!j.gather && out.empty() && threshold_applies(total, j.n, (double)j.n * vt::padded_dim(j.d) * 4.0)) {
    // one scan in key-column mode, exact threshold on the device, then the winners are
    // re-scored through the gather list for their raw values and sorted
    const uint32_t k = (uint32_t)total;
    VT_TRY(c.dKeyCol.ensure(((size_t)j.n + 1) / 2 * 2));
    VT_TRY(c.dPartKeys.ensure(kThresholdListCap));
    VT_TRY(c.dPartPay.ensure(kThresholdListCap));
    vt::ScanArgs a{};
    a.X = j.X;
    a.stride = j.stride;
    a.q = c.qsrc;
    a.id_rank = j.id_rank;
    a.n = j.n;
    a.d = j.d;
    a.metric = j.metric;
    a.order = j.order;
    a.k = 1;
    a.q_nonzero = j.q_nonzero;
    a.tile_rows = tile_rows;
    a.part_keys = c.dPartKeys.p;
    a.part_pay = c.dPartPay.p;
    a.status = c.dStatus.p;
    a.key_out = c.dKeyCol.p;
    const bool timed = c.profiling && count_profile;
    if (timed) VT_HIP(hipEventRecord(c.ev0, c.stream));
    VT_HIP(vt::launch_scan(a, c.grid_for(ntiles, vt::scan_lds_bytes(j.d, 1)), c.stream));
    if (timed) VT_HIP(hipEventRecord(c.ev1, c.stream));
    VT_TRY(threshold_rows(c, j.n, k));
    VT_TRY(c.dCandKeys.ensure(k));
    VT_TRY(c.dCandPay.ensure(k));
    vt::ScanArgs g = a;
    g.gather = &c.dListPay.p->row;
    g.gather_stride = sizeof(vt::Payload) / sizeof(uint32_t);
    g.n = k;
    g.tile_rows = 0;
    g.key_out = c.dCandKeys.p;
    g.pay_out = c.dCandPay.p;
    VT_HIP(vt::launch_scan(g, c.grid_for((k + vt::kTileRows - 1) / vt::kTileRows, vt::scan_lds_bytes(j.d, 1)), c.stream));
    const int rc = collect_sorted_list(c, c.dCandKeys.p, c.dCandPay.p, k, out);
    if (timed && rc != kRetryInternal) {
      float ms = 0.f;
      VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
      c.prof.scan_launches += 1;
      c.prof.scan_ms += ms;
      c.prof.scan_rows += j.n;
      c.prof.scan_bytes += (uint64_t)j.n * j.d * 4;
    }
    if (rc != kRetryInternal) return rc;
    out.clear();  // more equal keys at the threshold than the list holds: the pass-per-256 loop below
  }
  while (out.size() < total) {
    const uint32_t k = (uint32_t)std::min<size_t>(kmax, total - out.size());
    const uint32_t blocks = c.grid_for(ntiles, vt::scan_lds_bytes(j.d, k));
    const uint32_t waves = vt::scan_lists(blocks);
    VT_TRY(c.dPartKeys.ensure((size_t)waves * k));
    VT_TRY(c.dPartPay.ensure((size_t)waves * k));
    vt::ScanArgs a{};
    a.X = j.X;
    a.stride = j.stride;
    a.q = c.qsrc;
    a.id_rank = j.id_rank;
    a.gather = j.gather;
    a.gather_stride = j.gather_stride;
    a.n = j.n;
    a.d = j.d;
    a.metric = j.metric;
    a.order = j.order;
    a.k = k;
    a.lo_key = lo;
    a.has_lo = has_lo ? 1 : 0;
    a.q_nonzero = j.q_nonzero;
    a.tile_rows = tile_rows;
    a.part_keys = c.dPartKeys.p;
    a.part_pay = c.dPartPay.p;
    a.status = c.dStatus.p;
    const bool timed = c.profiling && count_profile;
    if (timed) VT_HIP(hipEventRecord(c.ev0, c.stream));
    VT_HIP(vt::launch_scan(a, blocks, c.stream));
    if (timed) VT_HIP(hipEventRecord(c.ev1, c.stream));
    VT_TRY(select_pass(c, c.dPartKeys.p, c.dPartPay.p, waves * k, k, 0, false));
    if (timed) {
      float ms = 0.f;
      VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
      c.prof.scan_launches += 1;
      c.prof.scan_ms += ms;
      c.prof.scan_rows += j.n;
      c.prof.scan_bytes += (uint64_t)j.n * j.d * 4;
      c.prof.merge_launches += 1;
    }
    if (c.hRes.p->status == vt::kStatusRetry) return kRetryInternal;
    if (c.hRes.p->status == VT_ERR_OVERFLOW) return VT_ERR_OVERFLOW;
    const uint32_t got = c.hRes.p->count;
    for (uint32_t i = 0; i < got; ++i) out.push_back(c.hRes.p->e[i]);
    if (got < k) break;
    lo = c.hRes.p->e[got - 1].key;
    has_lo = true;
  }
  return VT_OK;
}

// tile_pairs != 0: `d` is a prefix of rows whose bit tiles hold tile_pairs word pairs (HammingArgs).
int run_hamming(Ctx &c, const uint64_t *bits, const uint64_t *qbits, const uint32_t *id_rank, uint32_t n, uint32_t d,
                size_t want, std::vector<vt::Entry> &out, bool count_profile, bool jaccard = false, uint32_t tile_pairs = 0) {
  const uint32_t words = (d + 63) / 64;
  const uint32_t ntiles = (n + 63) / 64;
  uint64_t lo = 0;
  bool has_lo = false;
  const size_t total = std::min<size_t>(want, n);
  while (out.size() < total) {
    const uint32_t k = (uint32_t)std::min<size_t>((size_t)vt::kMaxFusedK, total - out.size());
    const uint32_t blocks = c.grid_for(ntiles, vt::hamming_lds_bytes(k), c.hamming_blocks_per_cu);
    const uint32_t waves = vt::scan_lists(blocks);
    VT_TRY(c.dPartKeys.ensure((size_t)waves * k));
    VT_TRY(c.dPartPay.ensure((size_t)waves * k));
    vt::HammingArgs a{};
    a.bits = bits;
    a.qbits = qbits;
    a.id_rank = id_rank;
    a.n = n;
    a.words = words;
    a.pairs = (words + 1) / 2;
    a.d = d;
    a.k = k;
    a.lo_key = lo;
    a.has_lo = has_lo ? 1 : 0;
    a.part_keys = c.dPartKeys.p;
    a.part_pay = c.dPartPay.p;
    a.jaccard = jaccard ? 1 : 0;
    a.tile_pairs = tile_pairs;
    if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
    VT_HIP(vt::launch_hamming(a, blocks, c.stream));
    if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
    VT_TRY(select_pass(c, c.dPartKeys.p, c.dPartPay.p, waves * k, k, 0, false));
    if (c.profiling && count_profile) {
      float ms = 0.f;
      VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
      c.prof.hamming_launches += 1;
      c.prof.hamming_ms += ms;
      c.prof.hamming_bytes += (uint64_t)n * words * 8;
    }
    const uint32_t got = c.hRes.p->count;
    for (uint32_t i = 0; i < got; ++i) out.push_back(c.hRes.p->e[i]);
    if (got < k) break;
    lo = c.hRes.p->e[got - 1].key;
    has_lo = true;
  }
  return VT_OK;
}

// Uploads a query of n floats into c.dQ padded with zeros to padded_dim(n).
// `with_bits`: the query's sign bits (compress_sign_bits, distances.rs:413-423: bit i % 64 of
// word i / 64 set iff v[i] >= 0.0, padding bits zero) are packed on the host -- n compares --
// and ride behind the floats in the same copy; c.dQbits points at them.
// with_bits == 2: the non-zero bits instead (bit set iff v[i] != 0.0: what float hamming / jaccard
// compare, distances.rs:319-347).
// (Reading the pinned block in place through its host mapping instead of copying it -- VT_DIRECT_QUERY in r05 -- takes a
// blit kernel and its launch out of the call and puts every block's first fetch on the link: a 4 096-row search 38.2 us
// in place against 35.7 copied, quantized 52.8 / 50.7, funnel 71.7 / 66.8, profiles/r05/latency_floor_*.jsonl.  It lost
// and has left the library: c.qsrc is always the device copy.)
int upload_query(Ctx &c, const float *q, size_t n, uint32_t *q_nonzero, int with_bits = 0) {
  const uint32_t ld = vt::padded_dim((uint32_t)n);
  const size_t words = ((n + 63) / 64 + 1) / 2 * 2;  // an even count (the odd one out zero): K4 / K4h read whole word pairs
  const size_t total = (size_t)ld + (with_bits ? 2 * words : 0);  // in floats (ld is a multiple of 64: the words are 8-byte aligned)
  VT_TRY(c.dQ.ensure(total));
  VT_TRY(c.hQ.ensure(total));
  std::memcpy(c.hQ.p, q, n * sizeof(float));
  for (size_t i = n; i < ld; ++i) c.hQ.p[i] = 0.0f;
  if (q_nonzero) {
    uint32_t nz = 0;
    for (size_t i = 0; i < n; ++i) nz += q[i] != 0.0f ? 1u : 0u;
    *q_nonzero = nz;
  }
  if (with_bits) {
    uint64_t *w = reinterpret_cast<uint64_t *>(c.hQ.p + ld);
    for (size_t i = 0; i < words; ++i) w[i] = 0;
    for (size_t i = 0; i < n; ++i)
      if (with_bits == 2 ? q[i] != 0.0f : q[i] >= 0.0f) w[i / 64] |= 1ull << (i % 64);
  }
  c.qbits_kind = with_bits;
  c.qsrc = c.dQ.p;
  VT_HIP(hipMemcpyAsync(c.dQ.p, c.hQ.p, total * sizeof(float), hipMemcpyHostToDevice, c.stream));
  if (with_bits) c.dQbits = reinterpret_cast<uint64_t *>(const_cast<float *>(c.qsrc) + ld);
  return VT_OK;
}

inline uint32_t rank_key_of(uint64_t key) { return (uint32_t)(key >> 32); }

}  // namespace
