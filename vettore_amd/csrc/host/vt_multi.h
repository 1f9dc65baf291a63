// vt_multi.h -- the handle level: shards of one vt_flat, workers, the exchange (RCCL / host), merges, staged searches across shards.
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// ======================================================= handle level (vt_flat)
int poisoned_status() { return fail(VT_ERR_POISONED, "flat lock poisoned"); }

constexpr int kStatusStaleRanks = 64;  // block status bit: the shard's externally installed id ranks no longer hold
constexpr size_t kExchangeBlockBytes = 16 + (size_t)vt::kMaxFusedK * sizeof(vt::Entry);

// Shard of an id: FNV-1a over the bytes, finished with a 64-bit mix (so that ids which
// differ in their last digits only still spread evenly).
inline uint32_t shard_of(const char *id, size_t len, size_t nshards) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < len; ++i) {
    h ^= (unsigned char)id[i];
    h *= 1099511628211ull;
  }
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 33;
  return (uint32_t)(h % nshards);
}

// Runs fn(s) for every shard s in `which` on that shard's worker thread, all at once;
// the status of the lowest failing shard wins (its detail text becomes this thread's).
template <class F>
int on_shards(vt_flat *h, const std::vector<size_t> &which, F fn) {
  return vt_host::run_on_workers(h->workers, h->post_mu, which, fn,
                                 [](int status, const std::string &error) { return fail(status, error); });
}
template <class F>
int on_all_shards(vt_flat *h, F fn) {
  std::vector<size_t> all(h->shards.size());
  for (size_t s = 0; s < all.size(); ++s) all[s] = s;
  return on_shards(h, all, fn);
}

// A read on a one-shard handle: under the shared lock on a leased context when the derived
// columns it needs are current; otherwise (or when a lazy search hit a tie only the true id
// order can cut) under the exclusive lock, which first brings them up to date.
template <class F>
int read_single(vt_flat *h, unsigned need, size_t limit, F &&fn) {
  Shard *ix = h->shards[0].get();
  bool escalated = false;
  {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    if (h->poisoned) return poisoned_status();
    if (!shard_stale(ix, need, limit)) {
      CtxLease lease(ix);
      if (!lease.c) return lease.status;
      VT_TRY(reader_ready(ix, *lease.c));
      const int st = fn(ix, *lease.c);
      if (st != kEscalate) return st;
      escalated = true;
    }
  }
  std::unique_lock<std::shared_mutex> wl(h->rw);
  if (h->poisoned) return poisoned_status();
  VT_TRY(ix->ctx.bind());
  if (escalated) need |= NEED_STRICT_RANKS;
  VT_TRY(shard_prepare(ix, need, limit));
  int st = fn(ix, ix->ctx);
  if (st == kEscalate) {
    // Under the exclusive lock nothing changes any more: whatever the body found stale a moment
    // ago (a writer may have pushed the corpus past a threshold between the caller's look at it
    // and this run, so that the body now wants norms it did not ask for) is brought up to date
    // here, all of it, and the body runs once more.  kEscalate never leaves this function.
    VT_TRY(shard_prepare(ix, need | NEED_STRICT_RANKS | NEED_NORMS, limit));
    st = fn(ix, ix->ctx);
    if (st == kEscalate) return fail(VT_ERR_DEVICE, "internal: a search found its derived columns stale under the exclusive lock");
  }
  return st;
}

// flat.rs:88-93 on one shard.
int shard_delete(Shard *ix, const char *id, size_t id_len, bool *began) {
  Ctx &c = ix->ctx;
  const char *idp = id ? id : "";
  const uint64_t hash = vt_host::hash_id(idp, id_len);
  const uint32_t found = ix->row_of.find(idp, id_len, hash);
  if (found != vt_host::IdTable::kNone) {
    *began = true;
    ix->epoch += 1;
    const uint32_t r = found, last = ix->n - 1;
    if (ix->rank_host[r] == kUnranked && ix->unranked) ix->unranked -= 1;
    ix->row_of.erase(idp, id_len, hash);
    if (r != last) {
      // (the table learns the last row's new place while ids[last] still holds its bytes)
      ix->row_of.move_row(ix->ids[last].data(), ix->ids[last].size(), vt_host::hash_id(ix->ids[last].data(), ix->ids[last].size()), r);
      // swap-delete: the last row moves into the hole and keeps its rank (the device side: one launch, below)
      ix->ids[r] = std::move(ix->ids[last]);
      ix->rank_host[r] = ix->rank_host[last];
      if (!ix->ranks_clean) {
        ix->rank_dirty.push_back(r);
        if (ix->rank_dirty.size() > kMaxDirtyRanks) ix->rank_dirty_all = true;
      }
    }
    VT_HIP(vt::launch_swap_delete(ix->dX, ix->ld, r, last, ix->ranks_clean && ix->dRank.p ? ix->dRank.p : nullptr, c.stream));
    // (a device-side move, queued on the primary stream: nothing to wait for here -- an event behind it is what
    // readers on other streams wait for, Shard::Landing; the slot's buffer goes unused)
    unsigned char *unused = nullptr;
    hipEvent_t ev = nullptr;
    VT_TRY(landing_slot(ix, &unused, &ev));
    VT_TRY(landing_record(ix, ev));
    ix->ids.pop_back();
    ix->rank_host.pop_back();
    ix->n -= 1;
    if (r != last) index_touch_row(ix, r);  // row r now holds what was the last row
  }
  if (ix->n == 0) {
    ix->dim = -1;
    ix->rank_dirty.clear();
    ix->rank_dirty_all = false;
    ix->unranked = 0;
    ix->ranks_clean = true;
    ix->max_id.clear();
    ix->max_rank = 0;
  }
  return VT_OK;
}

// Upload + scan + select into `device_block` ({i32 status, u32 count, pad[2]} then `limit`
// entries) on the context's stream, nothing waited for.  Ranks strictly current.
int shard_begin(Shard *ix, Ctx &c, const float *query, size_t n, size_t limit, void *device_block) {
  if (limit == 0 || limit > (size_t)vt::kMaxFusedK) return fail(VT_ERR_UNSUPPORTED, "search_begin needs 1 <= limit <= 256");
  VT_TRY(validate_vector(query, n, ix->dim));
  if (ix->n == 0) {
    VT_HIP(hipMemsetAsync(device_block, 0, 16, c.stream));  // count = 0
    return VT_OK;
  }
  uint32_t qnz = 0;
  VT_TRY(upload_query(c, query, n, &qnz));
  const uint32_t d = (uint32_t)ix->dim, k = (uint32_t)limit;
  if (vt::scan_lds_bytes(d, k) == 0) return fail(VT_ERR_UNSUPPORTED, "dimension/limit exceed the scan kernel's LDS");
  const uint32_t tile_rows = vt::scan_tile_rows(ix->n, d, c.resident_waves());
  const uint32_t blocks = c.grid_for((ix->n + tile_rows - 1) / tile_rows, vt::scan_lds_bytes(d, k));
  VT_TRY(c.dPartKeys.ensure((size_t)blocks * k));
  VT_TRY(c.dPartPay.ensure((size_t)blocks * k));
  vt::ScanArgs a{};
  a.X = ix->dX;
  a.stride = ix->ld;
  a.q = c.qsrc;
  a.id_rank = ix->dRank.p;
  a.n = ix->n;
  a.d = d;
  a.metric = ix->metric;
  a.order = ix->order;
  a.k = k;
  a.q_nonzero = qnz;
  a.tile_rows = tile_rows;
  a.part_keys = c.dPartKeys.p;
  a.part_pay = c.dPartPay.p;
  a.status = c.dStatus.p;
  if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
  VT_HIP(vt::launch_scan(a, blocks, c.stream));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
  c.begin_rows = ix->n;
  c.begin_dim = d;
  VT_HIP(vt::launch_select(c.dPartKeys.p, c.dPartPay.p, blocks * k, k, 0, 0, c.dStatus.p,
                           static_cast<ResultBlock *>(device_block), c.dSelKeys.p, c.dSelPay.p, c.stream));
  return VT_OK;
}

int settle_begin_profile(Ctx &c) {
  if (c.profiling && c.begin_rows) {
    float ms = 0.f;
    VT_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    c.prof.scan_launches += 1;
    c.prof.scan_ms += ms;
    c.prof.scan_rows += c.begin_rows;
    c.prof.scan_bytes += (uint64_t)c.begin_rows * c.begin_dim * 4;
  }
  c.begin_rows = 0;
  return VT_OK;
}

// ---- the shards' lists meet: merge by (rank key, id bytes) == FlatHit::cmp (flat.rs:34-40).
// Needs no global id ranks: within a shard the lists are already in that order, across
// shards the id bytes themselves decide.
struct MergeItem {
  uint32_t rank_key;
  float raw;
  const std::string *id;
};
inline bool merge_less(const MergeItem &a, const MergeItem &b) {
  if (a.rank_key != b.rank_key) return a.rank_key < b.rank_key;
  return *a.id < *b.id;
}
int merged_hits(std::vector<MergeItem> &items, size_t limit, vt_hits **out) {
  const size_t k = std::min(limit, items.size());
  std::partial_sort(items.begin(), items.begin() + k, items.end(), merge_less);
  auto h = std::make_unique<vt_hits>();
  h->ids.reserve(k);
  h->raw.reserve(k);
  h->rank_key.reserve(k);
  for (size_t i = 0; i < k; ++i) {
    h->ids.push_back(*items[i].id);
    h->raw.push_back(items[i].raw);
    h->rank_key.push_back(items[i].rank_key);
  }
  *out = h.release();
  return VT_OK;
}
int merge_hit_lists(const std::vector<vt_hits *> &lists, size_t limit, vt_hits **out) {
  std::vector<MergeItem> items;
  size_t total = 0;
  for (const vt_hits *l : lists)
    if (l) total += l->by_row_of ? l->rows.size() : l->ids.size();
  items.reserve(total);
  for (const vt_hits *l : lists) {
    if (!l) continue;
    if (l->by_row_of)  // (a shard's batch list: rows named, ids where they live)
      for (size_t i = 0; i < l->rows.size(); ++i) items.push_back(MergeItem{l->rank_key[i], l->raw[i], &l->by_row_of->ids[l->rows[i]]});
    else
      for (size_t i = 0; i < l->ids.size(); ++i) items.push_back(MergeItem{l->rank_key[i], l->raw[i], &l->ids[i]});
  }
  return merged_hits(items, limit, out);
}

// One communicator per shard (ncclCommInitAll: one process, all devices), the exchange
// blocks, and shard 0's pinned copy of the gathered lists.
int exchange_setup(vt_flat *h) {
  if (!h->comms.empty()) return VT_OK;
  const size_t S = h->shards.size();
  std::vector<int> devs(S);
  for (size_t s = 0; s < S; ++s) devs[s] = h->shards[s]->ctx.device;
  for (size_t a = 0; a < S; ++a)
    for (size_t b = a + 1; b < S; ++b)
      if (devs[a] == devs[b]) return fail(VT_ERR_UNSUPPORTED, "RCCL needs every shard on its own device");
  Rccl &r = rccl();
  if (!r.ok) return fail(VT_ERR_DEVICE, r.error);
  // buffers first, communicators last: the handle either has a complete exchange or none
  if (h->dBlock.empty()) {
    h->dBlock.assign(S, nullptr);
    h->dGather.assign(S, nullptr);
  }
  for (size_t s = 0; s < S; ++s) {
    VT_HIP(hipSetDevice(devs[s]));
    if (!h->dBlock[s]) VT_HIP(hipMalloc(&h->dBlock[s], kExchangeBlockBytes));
    if (!h->dGather[s]) VT_HIP(hipMalloc(&h->dGather[s], S * kExchangeBlockBytes));
  }
  VT_HIP(hipSetDevice(devs[0]));
  VT_TRY(h->hGather.ensure(S * kExchangeBlockBytes));
  std::vector<ncclComm_t> comms(S, nullptr);
  const ncclResult_t rc = r.CommInitAll(comms.data(), (int)S, devs.data());
  if (rc != ncclSuccess) return fail(VT_ERR_DEVICE, std::string("ncclCommInitAll: ") + r.GetErrorString(rc));
  h->comms = std::move(comms);
  return VT_OK;
}

// The wait behind a shard's all-gather, with a deadline.  A collective whose peers never arrive
// (a rank that died, a link that wedged) would otherwise block this worker -- and with it the
// caller, the handle and whoever drives the process -- until some outer timeout kills everything
// without a word.  VT_EXCHANGE_TIMEOUT_MS (default 20 000; a scan of a full 288-GB card takes 45 ms).
// On expiry the handle is poisoned: its streams still hold the stuck collective, nothing queued
// behind it will ever run.  A fresh process is the only retry.
int wait_exchange(vt_flat *h, Ctx &c, size_t shard) {
  const long timeout_ms = [] {
    const long v = vt::env::get(vt::env::EXCHANGE_TIMEOUT_MS);
    return v > 0 ? v : 20000L;
  }();
  // An event behind everything the shard has queued.  HIP has no timed wait, so the deadline is kept here -- but not by
  // spinning through the whole scan as r04 did (every in-flight search parked one worker per shard in a hipStreamQuery /
  // yield loop: eight cores burning for 2 ms per query at config 4's size, beside the BEAM's schedulers).  The context
  // remembers how long its last waits took: the worker SLEEPS through most of that (one sleep of 0.7 x the expected
  // time, when that is worth a timer at all) and only then polls, yielding between polls, so a search ends within
  // microseconds of its exchange -- a first form that napped 5..50 us between polls cost 30-50 us per query (a nap of
  // 5 us is 60 on this kernel; 0.615 -> 0.668 ms per step at 1.25 M rows per shard).
  if (!c.ev_wait) VT_HIP(hipEventCreateWithFlags(&c.ev_wait, hipEventDisableTiming));
  VT_HIP(hipEventRecord(c.ev_wait, c.stream));
  const auto t0 = std::chrono::steady_clock::now();
  auto waited_us = [&]() { return (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); };
  if (c.wait_expect_us > 400.0) std::this_thread::sleep_for(std::chrono::microseconds((long)(0.7 * c.wait_expect_us)));
  for (unsigned spins = 0;; ++spins) {
    const hipError_t e = hipEventQuery(c.ev_wait);
    if (e == hipSuccess) {
      c.wait_expect_us = 0.75 * c.wait_expect_us + 0.25 * (double)waited_us();
      return VT_OK;
    }
    if (e != hipErrorNotReady) {
      (void)hipGetLastError();
      return fail(VT_ERR_DEVICE, std::string("hipEventQuery behind the exchange: ") + hipGetErrorString(e));
    }
    if ((spins & 63u) == 63u) {
      const long w = waited_us();
      if (w >= timeout_ms * 1000) {
        h->poisoned = true;
        h->wedged = true;
        return fail(VT_ERR_DEVICE, "RCCL exchange timed out on shard " + std::to_string(shard) + " (device " +
                                       std::to_string(c.device) + ") after " + std::to_string(w / 1000) +
                                       " ms: the all-gather of the shards' top-k lists did not complete; the handle is unusable");
      }
      // far beyond what this context has ever waited (a peer that is late, not dead yet): stop burning the core
      if ((double)w > 4.0 * c.wait_expect_us + 2000.0) std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    std::this_thread::yield();
  }
}

#ifdef VT_TEST_HOOKS
// (libvettore_hip_hooks.so only) VT_TEST_EXCHANGE_STALL_MS=<ms>: the shard's stream waits that long
// on a flag before its all-gather -- what a peer that never arrives looks like from here
int test_stall_exchange(Ctx &c) {
  const long ms = vt::env::get(vt::env::TEST_EXCHANGE_STALL_MS);
  if (ms <= 0) return VT_OK;
  uint32_t *flag = nullptr, *dflag = nullptr;
  VT_HIP(hipHostMalloc(reinterpret_cast<void **>(&flag), sizeof(uint32_t), hipHostMallocMapped));
  *flag = 0;
  VT_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&dflag), flag, 0));
  VT_HIP(hipStreamWaitValue32(c.stream, dflag, 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
  std::thread([flag, ms] {
    std::this_thread::sleep_for(std::chrono::milliseconds(ms));
    __atomic_store_n(flag, 1u, __ATOMIC_RELEASE);  // (leaked on purpose: the stream may look at it any time after)
  }).detach();
  return VT_OK;
}
#endif

// flat_search on a multi-shard handle (shared lock held by the caller).
int search_multi(vt_flat *h, const float *query, size_t n, size_t limit, vt_hits **out) {
  if (limit == 0) return empty_hits(out);  // flat.rs:97-101: before the query is looked at
  VT_TRY(validate_vector(query, n, h->dim));
  if (h->total() == 0) return empty_hits(out);
  const size_t S = h->shards.size();
  const bool via_rccl = h->exchange == VT_EXCHANGE_RCCL && !h->comms.empty() && limit <= (size_t)vt::kMaxFusedK &&
                        vt::scan_lds_bytes((uint32_t)h->dim, (uint32_t)limit) != 0;
  if (via_rccl) {
    // every shard: scan + select into its device block, one all-gather queued behind them on
    // the shard's stream; shard 0 copies the gathered lists out; one wait per shard
    const size_t bytes = 16 + limit * sizeof(vt::Entry);
    Rccl &r = rccl();
    // (this call's copy of the gathered lists: hGather belongs to shard 0's worker, whose next job
    // -- another caller's search -- may refill it while this thread is still merging)
    std::vector<unsigned char> gathered;
    VT_TRY(on_all_shards(h, [&](size_t s) -> int {
      Shard *ix = h->shards[s].get();
      Ctx &c = ix->ctx;
      int st = VT_OK;
      if (shard_stale(ix, NEED_STRICT_RANKS, limit)) st = shard_prepare(ix, NEED_STRICT_RANKS, limit);
      if (st == VT_OK) st = shard_begin(ix, c, query, n, limit, h->dBlock[s]);
      if (st != VT_OK) {
        // the collective must still be entered by every shard: an empty block carrying the status
        uint32_t head[4] = {(uint32_t)st, 0, 0, 0};
        (void)hipMemcpyAsync(h->dBlock[s], head, sizeof head, hipMemcpyHostToDevice, c.stream);
        (void)hipStreamSynchronize(c.stream);
      }
#ifdef VT_TEST_HOOKS
      VT_TRY(test_stall_exchange(c));
#endif
      const ncclResult_t rc = r.AllGather(h->dBlock[s], h->dGather[s], bytes, ncclChar, h->comms[s], c.stream);
      if (rc != ncclSuccess) return fail(VT_ERR_DEVICE, std::string("ncclAllGather: ") + r.GetErrorString(rc));
      if (s == 0)
        VT_HIP(hipMemcpyAsync(h->hGather.p, h->dGather[0], S * bytes, hipMemcpyDeviceToHost, c.stream));
      VT_TRY(wait_exchange(h, c, s));
      if (s == 0) gathered.assign(h->hGather.p, h->hGather.p + S * bytes);
      VT_TRY(settle_begin_profile(c));
      return st;
    }));
    std::vector<MergeItem> items;
    for (size_t s = 0; s < S; ++s) {
      const unsigned char *blk = gathered.data() + s * bytes;
      int status;
      uint32_t count;
      std::memcpy(&status, blk, 4);
      std::memcpy(&count, blk + 4, 4);
      if (status == VT_ERR_OVERFLOW) return VT_ERR_OVERFLOW;
      if (status != VT_OK) return fail(VT_ERR_DEVICE, "a shard reported status " + std::to_string(status));
      const vt::Entry *e = reinterpret_cast<const vt::Entry *>(blk + 16);
      for (uint32_t i = 0; i < count && i < limit; ++i)
        items.push_back(MergeItem{rank_key_of(e[i].key), e[i].raw, &h->shards[s]->ids[e[i].row]});
    }
    return merged_hits(items, limit, out);
  }
  // host exchange: every shard's select kernel writes its list through the host mapping
  std::vector<vt_hits *> lists(S, nullptr);
  const int st = on_all_shards(h, [&](size_t s) -> int { return search_owner(h->shards[s].get(), query, n, limit, &lists[s]); });
  int rc = st;
  if (rc == VT_OK) rc = merge_hit_lists(lists, limit, out);
  for (vt_hits *l : lists) delete l;
  return rc;
}

// flat_search_batch on a multi-shard handle.
int batch_multi(vt_flat *h, const float *queries, size_t nq, size_t d, size_t limit, vt_hits **out) {
  if (limit == 0) {
    for (size_t i = 0; i < nq; ++i) VT_TRY(empty_hits(&out[i]));
    return VT_OK;
  }
  for (size_t i = 0; i < nq; ++i) VT_TRY(validate_vector(queries + i * d, d, h->dim));
  if (h->total() == 0) {
    for (size_t i = 0; i < nq; ++i) VT_TRY(empty_hits(&out[i]));
    return VT_OK;
  }
  const size_t S = h->shards.size();
  std::vector<std::vector<vt_hits *>> per(S, std::vector<vt_hits *>(nq, nullptr));
  // One flag per shard and query, set by the shard's worker (release) when that query's list is final -- group by group
  // on the matrix-core path (batch_ready's settle), everything that is left when the shard's job ends.  This thread
  // would only sleep until the last shard is through: it merges instead, query by query, whatever every shard has
  // settled.  A call of sixteen groups then ends one group's merge behind its last pass instead of sixteen (priced on
  // one GPU at config 4's shard size, 4 096 queries in one call: 41.4 -> 36.1 ms; per query the merge is S k
  // items, a partial sort and k id copies, serial and AFTER the passes before).
  std::unique_ptr<std::atomic<unsigned char>[]> fin(new std::atomic<unsigned char>[S * nq]);
  for (size_t i = 0; i < S * nq; ++i) fin[i].store(0, std::memory_order_relaxed);
  std::vector<char> merged(nq, 0);
  size_t first_open = 0;
  int merge_rc = VT_OK;
  std::vector<vt_hits *> lists(S);
  auto merge_ready = [&]() -> bool {
    if (merge_rc != VT_OK) return false;
    bool any = false;
    while (first_open < nq && merged[first_open]) ++first_open;
    // (lists settle group by group, in order, with a hole here and there -- a query the matrix cores could not certify,
    // settled when its shard ends: two groups' worth of unsettled queries in a row and the rest is not looked at this
    // time round.  A poll of a million-query call is not a walk over a million flags per shard.)
    size_t unsettled = 0;
    for (size_t i = first_open; i < nq && unsettled < 512; ++i) {
      if (merged[i]) continue;
      bool ready = true;
      for (size_t s = 0; s < S && ready; ++s) ready = fin[s * nq + i].load(std::memory_order_acquire) != 0;
      if (!ready) {
        unsettled += 1;
        continue;
      }
      for (size_t s = 0; s < S; ++s) lists[s] = per[s][i];
      // (nothing may unwind out of here: the workers' jobs point into this frame until the last of them has ended)
      merge_rc = no_throw([&]() -> int { return merge_hit_lists(lists, limit, &out[i]); });
      if (merge_rc != VT_OK) return false;
      for (size_t s = 0; s < S; ++s) {  // (final means the shard is through with it)
        delete per[s][i];
        per[s][i] = nullptr;
      }
      merged[i] = 1;
      any = true;
    }
    return any;
  };
  std::vector<size_t> all(S);
  for (size_t s = 0; s < S; ++s) all[s] = s;
  auto shard_job = [&](size_t s) -> int {
    Shard *ix = h->shards[s].get();
    const unsigned need = NEED_STRICT_RANKS | NEED_NZBITS | (batch_uses_mfma(ix, nq, limit) ? NEED_NORMS : 0u);
    if (shard_stale(ix, need, limit)) VT_TRY(shard_prepare(ix, need, limit));
    // (this worker's lists go into the merge and nowhere else: they name their rows, the ids -- 2 560 string
    // copies per 256 queries and shard -- are copied once, for the winners.  The request is this job's, on this thread.)
    const int st = guarded([&]() -> int {
      MergeRequestScope by_row(ix, fin.get() + s * nq);
      return batch_ready(ix, ix->ctx, queries, nq, d, limit, per[s].data());
    });
    if (st == VT_OK)
      for (size_t i = 0; i < nq; ++i) fin[s * nq + i].store(1, std::memory_order_release);
    return st;
  };
  // (one group or less: nothing is settled before the shards end, and a caller that polls hears of that end up to a
  // nap later than one that sleeps on the workers' condition variable)
  int rc = nq > 256 ? vt_host::run_on_workers_meanwhile(h->workers, h->post_mu, all, shard_job, merge_ready,
                                                        [](int status, const std::string &error) { return fail(status, error); })
                    : on_all_shards(h, shard_job);
  if (rc == VT_OK) rc = merge_rc;
  // What is left when the last shard is through: on the workers, idle by now, each its share (a quarter of a
  // millisecond per 256 queries at eight shards on one thread); a handful, here.
  std::vector<size_t> rest;
  for (size_t i = 0; i < nq && rc == VT_OK; ++i)
    if (!merged[i]) rest.push_back(i);
  if (rc == VT_OK && S > 1 && rest.size() >= 4 * S) {
    rc = on_all_shards(h, [&](size_t s) -> int {
      std::vector<vt_hits *> mine(S);
      for (size_t r = rest.size() * s / S; r < rest.size() * (s + 1) / S; ++r) {
        for (size_t t = 0; t < S; ++t) mine[t] = per[t][rest[r]];
        VT_TRY(merge_hit_lists(mine, limit, &out[rest[r]]));
      }
      return VT_OK;
    });
  } else {
    for (size_t r = 0; r < rest.size() && rc == VT_OK; ++r) {
      for (size_t s = 0; s < S; ++s) lists[s] = per[s][rest[r]];
      rc = merge_hit_lists(lists, limit, &out[rest[r]]);
    }
  }
  for (auto &v : per)
    for (vt_hits *l : v) delete l;
  return rc;
}

// ---- quantized / funnel / hybrid search on a multi-shard handle --------------------------
// Every step of those searches is "the best `keep` rows of a row set under some score"
// (binary_top_k, search.rs:76-92; vector_top_k on a prefix or on the full vectors,
// search.rs:38-73; the index's own search).  Each shard finds the best `keep` of ITS part of the
// row set, the handle merges the shards' lists by (rank key, id bytes) -- the order every one of
// those functions sorts by -- and the survivors are dealt back to their shards for the next
// step.  The global best `keep` are among the shards' best `keep`, so the result is the
// reference's.
enum StageKind { STAGE_HAMMING, STAGE_PREFIX, STAGE_SEARCH };
struct StageItem {
  uint32_t rank_key;
  float raw;
  uint32_t shard, row;
  const std::string *id;
};
using ShardRows = std::vector<std::vector<uint32_t>>;  // per shard: rows of the current candidate set

// `subset` == nullptr: all rows of every shard.  Keeps the best `keep`, ascending, in `out`.
int multi_stage(vt_flat *h, StageKind kind, uint32_t d, const float *query, size_t n, const ShardRows *subset, size_t keep,
                std::vector<StageItem> &out) {
  const size_t S = h->shards.size();
  std::vector<std::vector<vt::Entry>> per(S);
  std::vector<size_t> which;
  for (size_t s = 0; s < S; ++s)
    if (h->shards[s]->n && (!subset || !(*subset)[s].empty())) which.push_back(s);
  out.clear();
  if (which.empty()) return VT_OK;
  VT_TRY(on_shards(h, which, [&](size_t s) -> int {
    Shard *ix = h->shards[s].get();
    Ctx &c = ix->ctx;
    const unsigned need = NEED_STRICT_RANKS | (kind == STAGE_HAMMING ? NEED_BITS : 0u);
    if (shard_stale(ix, need, keep)) VT_TRY(shard_prepare(ix, need, keep));
    uint32_t qnz_full = 0;
    VT_TRY(upload_query(c, query, n, &qnz_full, kind == STAGE_HAMMING ? 1 : (kind == STAGE_PREFIX && pattern_metric(ix->metric)) ? 2 : 0));
    if (kind == STAGE_HAMMING) {
      std::vector<uint32_t> rows;
      return quantized_rows(ix, c, keep, rows, &per[s]);
    }
    if (kind == STAGE_SEARCH) {
      ScanJob j{};
      j.X = ix->dX;
      j.stride = ix->ld;
      j.id_rank = ix->dRank.p;
      j.n = ix->n;
      j.d = (uint32_t)ix->dim;
      j.metric = ix->metric;
      j.order = ix->order;
      j.q_nonzero = qnz_full;
      return run_scan(c, j, keep, per[s], false);
    }
    uint32_t nz = 0;
    for (uint32_t i = 0; i < d; ++i) nz += query[i] != 0.0f ? 1u : 0u;
    static const std::vector<uint32_t> none;
    return funnel_stage(ix, c, query, d, subset ? (*subset)[s] : none, subset == nullptr, keep, nz, per[s]);
  }));
  for (size_t s : which)
    for (const vt::Entry &e : per[s])
      out.push_back(StageItem{rank_key_of(e.key), e.raw, (uint32_t)s, e.row, &h->shards[s]->ids[e.row]});
  const size_t k = std::min(keep, out.size());
  auto less = [](const StageItem &a, const StageItem &b) {
    if (a.rank_key != b.rank_key) return a.rank_key < b.rank_key;
    return *a.id < *b.id;
  };
  std::partial_sort(out.begin(), out.begin() + k, out.end(), less);
  out.resize(k);
  return VT_OK;
}

void deal_rows(const std::vector<StageItem> &items, size_t nshards, ShardRows &rows) {
  rows.assign(nshards, {});
  for (const StageItem &it : items) rows[it.shard].push_back(it.row);
}

int stage_hits(const std::vector<StageItem> &items, vt_hits **out) {
  auto hh = std::make_unique<vt_hits>();
  for (const StageItem &it : items) {
    hh->ids.push_back(*it.id);
    hh->raw.push_back(it.raw);
    hh->rank_key.push_back(it.rank_key);
  }
  *out = hh.release();
  return VT_OK;
}

// funnel passes (collection.ex:674-691) without the final rerank: the candidate set per shard
int funnel_rows_multi(vt_flat *h, const float *query, size_t n, const size_t *stages, size_t nstages, size_t candidates,
                      ShardRows &rows, bool *empty) {
  std::vector<StageItem> kept;
  const ShardRows *subset = nullptr;
  for (size_t i = 0; i < nstages; ++i) {
    VT_TRY(multi_stage(h, STAGE_PREFIX, (uint32_t)stages[i], query, n, subset, candidates, kept));
    deal_rows(kept, h->shards.size(), rows);
    subset = &rows;
    if (kept.empty()) break;
  }
  *empty = kept.empty();
  return VT_OK;
}

// ---- the same searches in ONE round ------------------------------------------------------
// Only the first stage of a generator cuts the row set (every later funnel stage keeps the same
// `candidates`, collection.ex:674-691), and the handle-wide best `candidates` of that stage are
// among the shards' own best `candidates`.  So every shard runs its whole chain on its own
// candidates without waiting for anybody -- the chain a one-shard handle runs, the rerank left
// uncut -- and hands over (first-stage entries, rerank entries); the handle cuts the union of
// the first-stage lists to `candidates` by (rank key, id bytes), keeps the rerank entries of
// exactly those rows and orders them.  One fan-out instead of one per stage.  A shard that
// reports a metric overflow may have met it on a row the handle-wide set does not contain
// (the reference would not have looked at it): such a call is redone round by round, below.
bool staged_one_round() { return !vt::env::on(vt::env::STAGED_ROUNDS); }  // (tests force the round-per-stage path)

std::vector<size_t> shards_with_rows(const vt_flat *h) {
  std::vector<size_t> which;
  for (size_t s = 0; s < h->shards.size(); ++s)
    if (h->shards[s]->n) which.push_back(s);
  return which;
}

int staged_merge(vt_flat *h, const std::vector<size_t> &which, const std::vector<LocalStages> &loc,
                 const std::vector<size_t> &gen_keep, size_t limit, vt_hits **out) {
  auto less = [](const StageItem &a, const StageItem &b) {
    if (a.rank_key != b.rank_key) return a.rank_key < b.rank_key;
    return *a.id < *b.id;
  };
  std::unordered_set<uint64_t> chosen;
  std::vector<StageItem> items;
  for (size_t g = 0; g < gen_keep.size(); ++g) {
    items.clear();
    for (size_t s : which) {
      if (g >= loc[s].gens.size()) continue;
      for (const vt::Entry &e : loc[s].gens[g])
        items.push_back(StageItem{rank_key_of(e.key), e.raw, (uint32_t)s, e.row, &h->shards[s]->ids[e.row]});
    }
    const size_t k = std::min(gen_keep[g], items.size());
    std::partial_sort(items.begin(), items.begin() + k, items.end(), less);
    for (size_t i = 0; i < k; ++i) chosen.insert((uint64_t)items[i].shard << 32 | items[i].row);
  }
  items.clear();
  for (size_t s : which)
    for (const vt::Entry &e : loc[s].final_)
      if (chosen.count((uint64_t)s << 32 | e.row))
        items.push_back(StageItem{rank_key_of(e.key), e.raw, (uint32_t)s, e.row, &h->shards[s]->ids[e.row]});
  if (items.size() != chosen.size()) return kRetryInternal;  // a candidate without its rerank entry: not trusted
  const size_t k = std::min(limit, items.size());
  std::partial_sort(items.begin(), items.begin() + k, items.end(), less);
  items.resize(k);
  return stage_hits(items, out);
}

// fn(shard, context, &local) runs a shard's chain; returns true when *status is final.
template <class F>
bool staged_once(vt_flat *h, unsigned need, size_t prep_limit, const std::vector<size_t> &gen_keep, size_t limit,
                 vt_hits **out, int *status, F fn) {
  if (!staged_one_round()) return false;
  const std::vector<size_t> which = shards_with_rows(h);
  std::vector<LocalStages> loc(h->shards.size());
  int rc = on_shards(h, which, [&](size_t s) -> int {
    Shard *ix = h->shards[s].get();
    if (shard_stale(ix, need, prep_limit)) VT_TRY(shard_prepare(ix, need, prep_limit));
    return fn(ix, ix->ctx, &loc[s]);
  });
  if (rc == VT_OK) rc = staged_merge(h, which, loc, gen_keep, limit, out);
  if (rc == VT_ERR_OVERFLOW || rc == kRetryInternal) return false;
  *status = rc;
  return true;
}

int quantized_multi(vt_flat *h, const float *query, size_t n, size_t candidates, size_t limit, vt_hits **out) {
  VT_TRY(validate_vector(query, n, h->dim));  // collection.ex:276-295 via prepare_query
  if (h->total() == 0 || candidates == 0 || limit == 0) return empty_hits(out);
  int status = VT_OK;
  if (candidates <= (size_t)vt::kMaxFusedK &&
      staged_once(h, NEED_STRICT_RANKS | NEED_BITS, candidates, {candidates}, limit, out, &status,
                  [&](Shard *ix, Ctx &c, LocalStages *local) -> int {
                    vt_hits *none = nullptr;
                    return quantized_ready(ix, c, query, n, candidates, limit, &none, local);
                  }))
    return status;
  std::vector<StageItem> kept;
  VT_TRY(multi_stage(h, STAGE_HAMMING, 0, query, n, nullptr, candidates, kept));
  if (kept.empty()) return empty_hits(out);
  ShardRows rows;
  deal_rows(kept, h->shards.size(), rows);
  VT_TRY(multi_stage(h, STAGE_PREFIX, (uint32_t)h->dim, query, n, &rows, limit, kept));  // exact rerank, collection.ex:821-851
  return stage_hits(kept, out);
}

int funnel_multi(vt_flat *h, const float *query, size_t n, const size_t *stages, size_t nstages, size_t candidates,
                 size_t limit, vt_hits **out) {
  VT_TRY(validate_vector(query, n, h->dim));
  if (nstages == 0) return VT_ERR_PREFIX;
  for (size_t i = 0; i < nstages; ++i)
    if (stages[i] == 0 || stages[i] > n) return VT_ERR_PREFIX;
  if (h->total() == 0 || candidates == 0 || limit == 0) return empty_hits(out);
  int status = VT_OK;
  if (staged_once(h, NEED_STRICT_RANKS, candidates, {candidates}, limit, out, &status,
                  [&](Shard *ix, Ctx &c, LocalStages *local) -> int {
                    vt_hits *none = nullptr;
                    return funnel_ready(ix, c, query, n, stages, nstages, candidates, limit, &none, local);
                  }))
    return status;
  ShardRows rows;
  bool empty = false;
  VT_TRY(funnel_rows_multi(h, query, n, stages, nstages, candidates, rows, &empty));
  if (empty) return empty_hits(out);
  std::vector<StageItem> kept;
  VT_TRY(multi_stage(h, STAGE_PREFIX, (uint32_t)h->dim, query, n, &rows, limit, kept));
  return stage_hits(kept, out);
}

int hybrid_multi(vt_flat *h, const float *query, size_t n, const int *kinds, const size_t *candidates,
                 const size_t *stage_off, const size_t *stages, size_t ngen, size_t limit, vt_hits **out) {
  VT_TRY(validate_vector(query, n, h->dim));
  if (ngen == 0) return VT_ERR_ARGUMENT;
  for (size_t i = 0; i < ngen; ++i) {
    if (kinds[i] < VT_GEN_FUNNEL || kinds[i] > VT_GEN_SEARCH || candidates[i] == 0) return VT_ERR_ARGUMENT;
    if (kinds[i] == VT_GEN_FUNNEL) {
      if (stage_off[i + 1] <= stage_off[i]) return VT_ERR_PREFIX;
      for (size_t j = stage_off[i]; j < stage_off[i + 1]; ++j)
        if (stages[j] == 0 || stages[j] > n) return VT_ERR_PREFIX;
    }
  }
  if (h->total() == 0 || limit == 0) return empty_hits(out);
  {
    unsigned need = NEED_STRICT_RANKS;
    size_t most = limit;
    for (size_t i = 0; i < ngen; ++i) {
      if (kinds[i] == VT_GEN_QUANTIZED) need |= NEED_BITS;
      most = std::max(most, candidates[i]);
    }
    int status = VT_OK;
    if (staged_once(h, need, most, std::vector<size_t>(candidates, candidates + ngen), limit, out, &status,
                    [&](Shard *ix, Ctx &c, LocalStages *local) -> int {
                      vt_hits *none = nullptr;
                      return hybrid_ready(ix, c, query, n, kinds, candidates, stage_off, stages, ngen, limit, &none, local);
                    }))
      return status;
  }
  // hybrid_candidates (collection.ex:515-532): the union of the generators' candidate sets
  const size_t S = h->shards.size();
  ShardRows all(S), rows;
  std::vector<std::unordered_set<uint32_t>> seen(S);
  std::vector<StageItem> kept;
  for (size_t i = 0; i < ngen; ++i) {
    if (kinds[i] == VT_GEN_FUNNEL) {
      bool empty = false;
      VT_TRY(funnel_rows_multi(h, query, n, stages + stage_off[i], stage_off[i + 1] - stage_off[i], candidates[i], rows, &empty));
      if (empty) rows.assign(S, {});
    } else {
      VT_TRY(multi_stage(h, kinds[i] == VT_GEN_QUANTIZED ? STAGE_HAMMING : STAGE_SEARCH, 0, query, n, nullptr, candidates[i], kept));
      deal_rows(kept, S, rows);
    }
    for (size_t s = 0; s < S; ++s)
      for (uint32_t r : rows[s])
        if (seen[s].insert(r).second) all[s].push_back(r);
  }
  bool any = false;
  for (size_t s = 0; s < S; ++s) any = any || !all[s].empty();
  if (!any) return empty_hits(out);
  // hybrid_rerank :exact == exact_rerank (collection.ex:627-630, :821-851)
  VT_TRY(multi_stage(h, STAGE_PREFIX, (uint32_t)h->dim, query, n, &all, limit, kept));
  return stage_hits(kept, out);
}

// Shared body of insert_many / load_matrix / load_device_matrix once every row is validated:
// one shard takes the batch as it is; several shards take their rows (hash of the id) at
// the same time, each on its own worker.
int store_validated_rows(vt_flat *h, size_t count, const char *ids, const size_t *id_off, const RowSource &src, size_t d) {
  if (count == 0) return VT_OK;
  const size_t S = h->shards.size();
  if (!h->multi()) {
    Shard *ix = h->shards[0].get();
    if (ix->dim < 0) VT_TRY(index_set_dim(ix, d));
    bool began = false;
    const int st = no_throw([&]() -> int { return index_store_rows(ix, count, ids, id_off, src, &began); });
    if (st != VT_OK && began) h->poisoned = true;
    return st;
  }
  if (count > 0xFFFFFFF0ull) return fail(VT_ERR_UNSUPPORTED, "more than 2^32-16 rows in one batch");
  std::vector<std::vector<uint32_t>> pick(S);
  for (size_t i = 0; i < count; ++i) pick[shard_of(ids + id_off[i], id_off[i + 1] - id_off[i], S)].push_back((uint32_t)i);
  std::vector<size_t> which;
  for (size_t s = 0; s < S; ++s)
    if (!pick[s].empty()) which.push_back(s);
  std::vector<char> began(S, 0);
  const int st = on_shards(h, which, [&](size_t s) -> int {
    Shard *ix = h->shards[s].get();
    const std::vector<uint32_t> &mine = pick[s];
    // this shard's ids, packed
    std::vector<size_t> off(mine.size() + 1, 0);
    for (size_t i = 0; i < mine.size(); ++i) off[i + 1] = off[i] + (id_off[mine[i] + 1] - id_off[mine[i]]);
    std::string blob;
    blob.resize(off.back());
    for (size_t i = 0; i < mine.size(); ++i)
      std::memcpy(&blob[off[i]], ids + id_off[mine[i]], off[i + 1] - off[i]);
    if (ix->dim < 0) VT_TRY(index_set_dim(ix, d));
    RowSource sub = src;
    sub.pick = mine.data();
    bool b = false;
    const int r = no_throw([&]() -> int { return index_store_rows(ix, mine.size(), blob.data(), off.data(), sub, &b); });
    began[s] = b ? 1 : 0;
    return r;
  });
  if (st != VT_OK)
    for (size_t s = 0; s < S; ++s)
      if (began[s]) h->poisoned = true;
  if (st == VT_OK) h->dim = (long)d;
  return st;
}

long handle_dim(const vt_flat *h) { return h->multi() ? h->dim : h->shards[0]->dim; }

void refresh_approx_bytes(vt_flat *h) {
  uint64_t b = 0, r = 0;
  for (auto &sh : h->shards) {
    b += (uint64_t)sh->n * sh->ld * sizeof(float);
    r += sh->n;
  }
  h->approx_bytes.store(b, std::memory_order_relaxed);
  h->approx_rows.store(r, std::memory_order_relaxed);
}

int store_validated(vt_flat *h, size_t count, const char *ids, const size_t *id_off, const RowSource &src, size_t d) {
  const int st = store_validated_rows(h, count, ids, id_off, src, d);
  refresh_approx_bytes(h);
  return st;
}

// flat_search as one caller runs it (nifs.rs:297-309).
int search_direct(vt_flat *h, const float *query, size_t n, size_t limit, vt_hits **out) {
  if (h->multi()) {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    if (h->poisoned) return poisoned_status();
    return search_multi(h, query, n, limit, out);
  }
  // (vt_flat_set_single_nominate: the first lone search then also brings the norms and the bf16 shadow up to date)
  const Shard *s0 = h->shards[0].get();
  const unsigned need = NEED_RANKS | NEED_NZBITS |
                        (s0->single_nominate && limit <= (size_t)vt::kMaxFusedK ? NEED_NORMS | NEED_STRICT_RANKS : 0u);
  return read_single(h, need, limit, [&](Shard *ix, Ctx &c) -> int { return search_ready(ix, c, query, n, limit, out); });
}

// quantized_search as one caller runs it (collection.ex:276-295).
int quantized_direct(vt_flat *h, const float *query, size_t n, size_t candidates, size_t limit, vt_hits **out) {
  if (h->multi()) {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    if (h->poisoned) return poisoned_status();
    return quantized_multi(h, query, n, candidates, limit, out);
  }
  return read_single(h, NEED_STRICT_RANKS | NEED_BITS, limit, [&](Shard *ix, Ctx &c) -> int {
    return quantized_ready(ix, c, query, n, candidates, limit, out);
  });
}

// nq quantized searches, one hit list each (or one status for all): groups of up to eight share
// a sweep of the bit matrix on a one-shard handle; a multi-shard handle takes them one by one.
int quantized_batch_direct(vt_flat *h, const float *queries, size_t nq, size_t d, size_t candidates, size_t limit, vt_hits **out) {
  for (size_t i = 0; i < nq; ++i) out[i] = nullptr;
  int st = VT_OK;
  if (h->multi()) {
    for (size_t i = 0; i < nq && st == VT_OK; ++i) st = quantized_direct(h, queries + i * d, d, candidates, limit, &out[i]);
  } else {
    st = read_single(h, NEED_STRICT_RANKS | NEED_BITS, limit, [&](Shard *ix, Ctx &c) -> int {
      for (size_t i = 0; i < nq; ++i) {  // (a second run after an escalation starts clean)
        delete out[i];
        out[i] = nullptr;
      }
      return quantized_batch_ready(ix, c, queries, nq, d, candidates, limit, out);
    });
  }
  if (st != VT_OK)
    for (size_t i = 0; i < nq; ++i) {
      delete out[i];
      out[i] = nullptr;
    }
  return st;
}

// funnel_search on any handle.
int funnel_direct(vt_flat *h, const float *query, size_t n, const size_t *stages, size_t nstages, size_t candidates,
                  size_t limit, vt_hits **out) {
  *out = nullptr;
  if (h->multi()) {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    if (h->poisoned) return poisoned_status();
    return funnel_multi(h, query, n, stages, nstages, candidates, limit, out);
  }
  // (NEED_NZBITS: on a float hamming / jaccard collection the first stage reads the non-zero-bit column)
  return read_single(h, NEED_STRICT_RANKS | NEED_NZBITS, std::min<size_t>(candidates, vt::kMaxFusedK), [&](Shard *ix, Ctx &c) -> int {
    return funnel_ready(ix, c, query, n, stages, nstages, candidates, limit, out);
  });
}

// funnel_search for nq queries with one set of stages (vt_flat_funnel_search_batch).
int funnel_batch_direct(vt_flat *h, const float *queries, size_t nq, size_t d, const size_t *stages, size_t nstages,
                        size_t candidates, size_t limit, vt_hits **out) {
  for (size_t i = 0; i < nq; ++i) out[i] = nullptr;
  int st = VT_OK;
  if (h->multi()) {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    if (h->poisoned) return poisoned_status();
    for (size_t i = 0; i < nq && st == VT_OK; ++i) st = funnel_multi(h, queries + i * d, d, stages, nstages, candidates, limit, &out[i]);
  } else {
    // (NEED_NZBITS: as funnel_direct -- the singles of a float hamming / jaccard batch read the bit column)
    st = read_single(h, NEED_STRICT_RANKS | NEED_NZBITS, std::min<size_t>(candidates, vt::kMaxFusedK), [&](Shard *ix, Ctx &c) -> int {
      for (size_t i = 0; i < nq; ++i) {  // (a second run after an escalation starts clean)
        delete out[i];
        out[i] = nullptr;
      }
      return funnel_batch_ready(ix, c, queries, nq, d, stages, nstages, candidates, limit, out);
    });
  }
  if (st != VT_OK)
    for (size_t i = 0; i < nq; ++i) {
      delete out[i];
      out[i] = nullptr;
    }
  return st;
}

// flat_search_batch: nq queries of d floats, one hit list each, or one status for all.
int batch_direct(vt_flat *h, const float *queries, size_t nq, size_t d, size_t limit, vt_hits **out) {
  for (size_t i = 0; i < nq; ++i) out[i] = nullptr;
  int st;
  if (h->multi()) {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    if (h->poisoned) return poisoned_status();
    st = batch_multi(h, queries, nq, d, limit, out);
  } else {
    unsigned need = NEED_STRICT_RANKS | NEED_NZBITS;
    {
      std::shared_lock<std::shared_mutex> rl(h->rw);
      if (batch_uses_mfma(h->shards[0].get(), nq, limit)) need |= NEED_NORMS;
    }
    st = read_single(h, need, limit, [&](Shard *ix, Ctx &c) -> int {
      for (size_t i = 0; i < nq; ++i) {  // (a second run after an escalation starts clean)
        delete out[i];
        out[i] = nullptr;
      }
      if (ix->n && batch_uses_mfma(ix, nq, limit) && shard_stale(ix, NEED_NORMS, limit)) return kEscalate;
      return batch_ready(ix, c, queries, nq, d, limit, out);
    });
  }
  if (st != VT_OK)
    for (size_t i = 0; i < nq; ++i) {
      delete out[i];
      out[i] = nullptr;
    }
  return st;
}

}  // namespace
