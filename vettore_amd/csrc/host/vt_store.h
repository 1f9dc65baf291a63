// vt_store.h -- index ops on one shard: growing the slab, the id table and its ranks, storing rows.
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// ------------------------------------------------------------------ index ops
int index_reserve_once(Shard *ix, uint32_t want_rows);
// The rows come first: when the slab cannot grow and the shard keeps a bf16 shadow (an accelerator
// of the batch pass, half the slab's size), the shadow goes and the growth is tried once more.
int index_reserve(Shard *ix, uint32_t want_rows) {
  int st = index_reserve_once(ix, want_rows);
  if (st != VT_OK && st != VT_ERR_UNSUPPORTED && ix->dShadow.p) {
    (void)hipStreamSynchronize(ix->ctx.stream);
    ix->dShadow.release();
    ix->sh_valid = false;
    ix->sh_refused = true;
    ix->sh_dirty.clear();
    st = index_reserve_once(ix, want_rows);
  }
  return st;
}

int index_reserve_once(Shard *ix, uint32_t want_rows) {
  if (want_rows <= ix->cap) return VT_OK;
  const size_t row_bytes = (size_t)ix->ld * sizeof(float);
  auto tiles_up = [](uint64_t rows) { return (rows + vt::kTileRows - 1) / vt::kTileRows * vt::kTileRows; };
  if (tiles_up(want_rows) > 0xFFFFFFF0ull) return fail(VT_ERR_UNSUPPORTED, "more than 2^32-16 rows");
  Slab &sl = ix->slab;
  hipStream_t stream = ix->ctx.stream;
  const size_t old_bytes = (size_t)ix->n * row_bytes;
  const size_t need = (size_t)tiles_up(want_rows) * row_bytes;
  if (sl.mapped) {
    // more chunks behind the ones in use: the rows stay where they are
    VT_TRY(sl.map_up_to(need, ix->ctx.device));
    // rows n..cap are scanned by the last tile: keep them defined
    VT_HIP(hipMemsetAsync(reinterpret_cast<char *>(sl.p) + sl.defined, 0, sl.bytes - sl.defined, stream));
    VT_HIP(hipStreamSynchronize(stream));
    sl.defined = sl.bytes;
  } else {
    Slab fresh;
    bool have = false;
    if (need >= Slab::chunk_bytes() && Slab::mapping_allowed()) {
      const int st = fresh.start_mapped(need, ix->ctx.device);
      if (st == VT_OK) have = true;
      else if (st != VT_ERR_UNSUPPORTED) return st;
    }
    if (!have) {
      uint64_t nc = std::max<uint64_t>(want_rows, (uint64_t)ix->cap * 2);
      nc = tiles_up(std::max<uint64_t>(nc, 1024));
      // (doubling stops at one chunk: the growth after that maps instead of copying)
      if (Slab::mapping_allowed() && nc * row_bytes > Slab::chunk_bytes())
        nc = std::max<uint64_t>(tiles_up(want_rows), Slab::chunk_bytes() / row_bytes / vt::kTileRows * vt::kTileRows);
      if (nc > 0xFFFFFFF0ull) nc = tiles_up(want_rows);
      VT_HIP(hipMalloc(reinterpret_cast<void **>(&fresh.p), (size_t)nc * row_bytes));
      fresh.bytes = (size_t)nc * row_bytes;
    }
    if (sl.p && ix->n) VT_HIP(hipMemcpyAsync(fresh.p, sl.p, old_bytes, hipMemcpyDeviceToDevice, stream));
    VT_HIP(hipMemsetAsync(reinterpret_cast<char *>(fresh.p) + old_bytes, 0, fresh.bytes - old_bytes, stream));
    VT_HIP(hipStreamSynchronize(stream));
    sl.release();
    sl.p = fresh.p;
    sl.bytes = sl.defined = fresh.bytes;
    sl.mapped = fresh.mapped;
    sl.reserved = fresh.reserved;
    sl.chunk = fresh.chunk;
    sl.chunks.swap(fresh.chunks);
    fresh.p = nullptr;  // (ownership moved)
    fresh.bytes = fresh.reserved = 0;
    fresh.mapped = false;
  }
  ix->dX = sl.p;
  const uint64_t rows = std::min<uint64_t>(sl.bytes / row_bytes / vt::kTileRows * vt::kTileRows, 0xFFFFFFE0ull);
  ix->cap = (uint32_t)rows;
  return VT_OK;
}

// Sets the dimension of an empty index (first insert after creation/emptying).
int index_set_dim(Shard *ix, size_t d) {
  if (d > 0x7fffffffu) return fail(VT_ERR_UNSUPPORTED, "dimension too large");
  if (vt::scan_lds_bytes((uint32_t)d, 1) == 0)
    return fail(VT_ERR_UNSUPPORTED, "dimension " + std::to_string(d) + " exceeds what the scan kernel stages in LDS");
  const uint32_t ld = vt::padded_dim((uint32_t)d);
  ix->for_each_ctx([](Ctx &c) { c.ham_dirty = true; });  // K4h's histograms are cleared for d + 1 bins only: a new dimension starts clean
  ix->bits_valid = false;    // derived per-row data belongs to the old rows
  ix->nz_valid = false;
  ix->nz_refused = false;
  ix->max_sqnorm = -1.0;
  ix->bits_dirty.clear();
  ix->nz_dirty.clear();
  ix->norm_dirty.clear();
  ix->sh_valid = false;
  ix->sh_refused = false;
  ix->sh_dirty.clear();
  ix->dShadow.release();  // (an emptied index gives the room back; the next batch builds anew)
  if (ld != ix->ld) {
    VT_HIP(hipStreamSynchronize(ix->ctx.stream));
    ix->slab.release();
    ix->dX = nullptr;
    ix->cap = 0;
    ix->ld = ld;
  }
  ix->dim = (long)d;
  return VT_OK;
}

// Row for `id`: existing row, or a fresh one appended (ids/rank bookkeeping).
uint32_t index_row_for(Shard *ix, const char *id, size_t len, bool *is_new, uint64_t hash) {
  const uint32_t have = ix->row_of.find(id, len, hash);
  if (have != vt_host::IdTable::kNone) {
    *is_new = false;
    return have;
  }
  std::string key(id, len);
  ix->row_of.insert(hash, ix->n);  // (its bytes arrive in ids[n] at the end of this function)
  const uint32_t r = ix->n++;
  *is_new = true;
  ix->epoch += 1;
  if (ix->external_ranks) {
    // externally supplied ranks describe the old row set only: fall back to a local re-rank
    ix->external_ranks = false;
    ix->ranks_clean = false;
    std::fill(ix->rank_host.begin(), ix->rank_host.end(), kUnranked);
    ix->rank_dirty_all = true;  // the whole device column is stale now
    ix->unranked = ix->rank_host.size();
  }
  if (ix->ranks_clean) {
    // ids arriving in ascending order (snapshot rebuild sorts by id,
    // collection.ex:427-433) keep ranks valid without a re-sort
    if (r == 0 || id_less(ix->max_id, key)) {
      const uint32_t rk = r == 0 ? 0 : ix->max_rank + 1;
      if (r != 0 && ix->max_rank >= kUnranked - 1) ix->ranks_clean = false;
      ix->rank_host.push_back(rk);
      ix->max_rank = rk;
      ix->max_id = key;
    } else {
      ix->ranks_clean = false;
      ix->rank_host.push_back(kUnranked);
      ix->unranked += 1;
    }
  } else {
    ix->rank_host.push_back(kUnranked);
    ix->unranked += 1;
  }
  ix->ids.push_back(std::move(key));
  return r;
}

// Recomputes id_rank (position of each row's id in bytewise order) if stale
// and makes the device copy current.
// host_only: the ranking itself (ranks_clean afterwards), the device column left for the caller to bring up to date.
// fresh_sorted: the unranked rows already in id order (a bulk load sorts its ids while the rows travel), or null.
int index_sync_ranks(Shard *ix, bool force_upload, bool host_only = false, const std::vector<uint32_t> *fresh_sorted = nullptr) {
  if (!ix->ranks_clean) {
    // Rows that kept a rank from before are still in the right relative order
    // (ranks only need to be order-isomorphic to the ids): sort them by rank
    // (integers), sort only the unranked newcomers by id (strings), and merge.
    const std::vector<std::string> &ids = ix->ids;
    const std::vector<uint32_t> &rk = ix->rank_host;
    std::vector<uint32_t> ranked, fresh_own;
    uint32_t maxr = 0;
    size_t nranked = 0;
    {
      // (a bulk load of ten million unsorted ids comes through here once: the passes over the column are shared out)
      std::mutex mu;
      parallel_for(ix->n, [&](size_t lo, size_t hi) {
        uint32_t m = 0;
        size_t c = 0;
        for (size_t i = lo; i < hi; ++i)
          if (rk[i] != kUnranked) {
            m = std::max(m, rk[i]);
            ++c;
          }
        std::lock_guard<std::mutex> g(mu);
        maxr = std::max(maxr, m);
        nranked += c;
      });
    }
    // the caller's newcomers, already in id order, are taken as they are when they are ALL the unranked rows
    const bool given = fresh_sorted && fresh_sorted->size() == (size_t)ix->n - nranked;
    if (nranked && (uint64_t)maxr < 4ull * ix->n + 1024) {
      // ranks are unique: a bucket pass puts the ranked rows in rank (= id) order without sorting
      std::vector<uint32_t> slot((size_t)maxr + 1, kUnranked);
      parallel_for(ix->n, [&](size_t lo, size_t hi) {  // (ranks are unique: every slot has one writer)
        for (size_t i = lo; i < hi; ++i)
          if (rk[i] != kUnranked) slot[rk[i]] = (uint32_t)i;
      });
      ranked.reserve(nranked);
      for (uint32_t v : slot)
        if (v != kUnranked) ranked.push_back(v);
      if (!given) {
        fresh_own.reserve((size_t)ix->n - nranked);
        for (uint32_t i = 0; i < ix->n; ++i)
          if (rk[i] == kUnranked) fresh_own.push_back(i);
      }
    } else if (nranked == 0 && given) {
      // (nothing ranked yet: no pass at all)
    } else {
      ranked.reserve(nranked);
      for (uint32_t i = 0; i < ix->n; ++i) {
        if (rk[i] != kUnranked) ranked.push_back(i);
        else if (!given) fresh_own.push_back(i);
      }
      parallel_sort(ranked, [&rk](uint32_t a, uint32_t b) { return rk[a] < rk[b]; });
    }
    if (!given) parallel_sort(fresh_own, [&ids](uint32_t a, uint32_t b) { return ids[a] < ids[b]; });
    const std::vector<uint32_t> &fresh = given ? *fresh_sorted : fresh_own;
    std::vector<uint32_t> order(ix->n);
    if (fresh.size() < ranked.size() / 16) {
      // few newcomers: each finds its place among the ranked rows by binary search (string
      // compares only there), the merge itself moves integers
      std::vector<uint32_t> pos(fresh.size());
      parallel_for(fresh.size(), [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i)
          pos[i] = (uint32_t)(std::lower_bound(ranked.begin(), ranked.end(), fresh[i],
                                               [&ids](uint32_t a, uint32_t b) { return ids[a] < ids[b]; }) - ranked.begin());
      });
      size_t o = 0, f = 0;
      for (size_t r = 0; r <= ranked.size(); ++r) {
        while (f < fresh.size() && pos[f] == r) order[o++] = fresh[f++];
        if (r < ranked.size()) order[o++] = ranked[r];
      }
    } else if (ranked.size() < fresh.size() / 16) {
      // the other way round (a bulk load of unsorted ids onto a small or empty index: a handful of ranked rows, millions
      // of sorted newcomers): the few find their places among the many, the merge moves integers -- a plain merge
      // compared two id strings per newcomer, 0.1 s per ten million
      std::vector<size_t> pos(ranked.size());
      for (size_t i = 0; i < ranked.size(); ++i)
        pos[i] = (size_t)(std::lower_bound(fresh.begin(), fresh.end(), ranked[i],
                                           [&ids](uint32_t a, uint32_t b) { return ids[a] < ids[b]; }) - fresh.begin());
      size_t o = 0, r = 0;
      for (size_t f = 0; f <= fresh.size(); ++f) {
        while (r < ranked.size() && pos[r] == f) order[o++] = ranked[r++];
        if (f < fresh.size()) order[o++] = fresh[f];
      }
    } else {
      parallel_merge(ranked, fresh, order, [&ids](uint32_t a, uint32_t b) { return ids[a] < ids[b]; });
    }
    ix->rank_host.resize(ix->n);
    parallel_for(ix->n, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) ix->rank_host[order[i]] = (uint32_t)i;
    });
    if (ix->n) {
      ix->max_id = ids[order[ix->n - 1]];
      ix->max_rank = ix->n - 1;
    }
    ix->ranks_clean = true;
    ix->unranked = 0;
    force_upload = true;
    if (host_only) ix->rank_dirty_all = true;
  }
  if (host_only) return VT_OK;
  if (force_upload) {
    ix->rank_dirty.clear();
    ix->rank_dirty_all = false;
  }
  if (force_upload && ix->n) {
    VT_TRY(ix->dRank.ensure(std::max<size_t>(ix->cap, ix->n)));
    VT_HIP(hipMemcpyAsync(ix->dRank.p, ix->rank_host.data(), (size_t)ix->n * sizeof(uint32_t), hipMemcpyHostToDevice,
                          ix->ctx.stream));
    VT_HIP(hipStreamSynchronize(ix->ctx.stream));
  }
  return VT_OK;
}

// Brings the device rank column in line with rank_host WITHOUT re-ranking: newcomers keep
// kUnranked (all equal), which is enough for a search whose k-th and (k+1)-th hits differ in
// their f32 rank (search_locked checks exactly that and orders equal ranks by id bytes on
// the host).  An unsorted insert therefore costs the next search a few bytes, not an O(n)
// merge and a column upload.
int index_lazy_ranks(Shard *ix) {
  Ctx &c = ix->ctx;
  if (ix->n == 0) return VT_OK;
  if (ix->dRank.count < std::max<size_t>(ix->cap, ix->n)) {
    VT_TRY(ix->dRank.ensure(std::max<size_t>(ix->cap, ix->n)));
    ix->rank_dirty_all = true;
  }
  if (ix->rank_dirty_all) {
    VT_HIP(hipMemcpyAsync(ix->dRank.p, ix->rank_host.data(), (size_t)ix->n * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
    VT_HIP(hipStreamSynchronize(c.stream));
  } else if (!ix->rank_dirty.empty()) {
    const size_t m = ix->rank_dirty.size();
    VT_TRY(c.hRankPairs.ensure(2 * m));
    VT_TRY(c.dRankPairs.ensure(2 * m));
    size_t live = 0;
    for (uint32_t r : ix->rank_dirty) {
      if (r >= ix->n) continue;  // deleted since
      c.hRankPairs.p[2 * live] = r;
      c.hRankPairs.p[2 * live + 1] = ix->rank_host[r];
      ++live;
    }
    if (live) {
      VT_HIP(hipMemcpyAsync(c.dRankPairs.p, c.hRankPairs.p, 2 * live * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
      VT_HIP(vt::launch_scatter_u32(c.dRankPairs.p, (uint32_t)live, ix->dRank.p, c.stream));
    }
  }
  ix->rank_dirty.clear();
  ix->rank_dirty_all = false;
  return VT_OK;
}

// Device a pointer lives on (-1: not device memory we can tell).
int device_of_pointer(const void *p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return attr.device;
}

struct RowSource {
  const float *host = nullptr;    // host rows (ragged or dense)
  const size_t *off = nullptr;    // ragged offsets; null => dense with `d`
  const float *device = nullptr;  // dense device matrix [count][d]
  size_t d = 0;
  const uint32_t *pick = nullptr;  // optional: row i of this batch is row pick[i] of the source
  // dense host rows whose finiteness has NOT been checked yet: a bulk store checks them itself, beside
  // the copy to the device (index_store_bulk_host) -- anything else must be handed validated rows
  bool unvalidated = false;
};

constexpr size_t kMaxDerivedDirty = 65536;  // more mutated rows than this: rebuild instead of patching

// Row `r` changed: its sign bits, non-zero bits and norm are stale.
inline void index_touch_row(Shard *ix, uint32_t r) {
  if (ix->bits_valid) {
    ix->bits_dirty.push_back(r);
    if (ix->bits_dirty.size() > kMaxDerivedDirty) {
      ix->bits_valid = false;
      ix->bits_dirty.clear();
    }
  }
  if (ix->nz_valid) {
    ix->nz_dirty.push_back(r);
    if (ix->nz_dirty.size() > kMaxDerivedDirty) {
      ix->nz_valid = false;
      ix->nz_dirty.clear();
    }
  }
  if (ix->max_sqnorm >= 0.0) {
    ix->norm_dirty.push_back(r);
    if (ix->norm_dirty.size() > kMaxDerivedDirty) {
      ix->max_sqnorm = -1.0;
      ix->norm_dirty.clear();
    }
  }
  if (ix->sh_valid) {
    ix->sh_dirty.push_back(r);
    if (ix->sh_dirty.size() > kMaxDerivedDirty) {
      ix->sh_valid = false;
      ix->sh_dirty.clear();
    }
  }
}

// Uploads a row list (rows still < n) for the patch kernels; returns its length.
int upload_row_list(Shard *ix, std::vector<uint32_t> &list, uint32_t *count) {
  Ctx &c = ix->ctx;
  std::sort(list.begin(), list.end());
  list.erase(std::unique(list.begin(), list.end()), list.end());
  while (!list.empty() && list.back() >= ix->n) list.pop_back();
  *count = (uint32_t)list.size();
  if (list.empty()) return VT_OK;
  VT_TRY(c.hRankPairs.ensure(list.size()));
  VT_TRY(c.dRankPairs.ensure(list.size()));
  std::memcpy(c.hRankPairs.p, list.data(), list.size() * sizeof(uint32_t));
  VT_HIP(hipMemcpyAsync(c.dRankPairs.p, c.hRankPairs.p, list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
  return VT_OK;
}

// ---- a dense host matrix of new rows, all at once (vt_flat_load_matrix: the snapshot rebuild of
// collection.ex:427-433, a first load) ------------------------------------------------------------
// r03 ran the phases one after the other -- finiteness check, id table, rows to the device, id ranks:
// 0.14 + 0.30 + 0.15 s per 4 M rows, the link idle half the time (18 GB/s).  Here they overlap:
//   * the rows start for the device at once, through four pinned quarters whose copies alternate between two
//     streams, into the FREE rows behind the index (row n + i for batch row i: where they belong if every id is
//     new, the normal case);
//   * the finiteness check (flat.rs:69-85: the whole batch before anything is stored) rides on that copy: the
//     threads that fill a quarter look at every row they copy, and publish how far the batch is verified;
//   * the id table (one thread: hashes formed beforehand on several, then an insert per id) follows that mark,
//     and the ranking of the ids runs behind it, while the link is still busy;
//   * a row that fails the check after ids went in: they come out again (rollback in the id thread), the rows
//     that reached the slab's free space are zeroed again, the call fails as a whole and the index is what it was.
// Ids that turn out not to be all new and distinct (upserts, duplicates in the batch) send the batch
// through the general path below after all (the free rows zeroed first).
constexpr int kRetryGeneral = -201;
int index_store_rows(Shard *ix, size_t count, const char *ids, const size_t *id_off, const RowSource &src, bool *began);

int index_store_bulk_host(Shard *ix, size_t count, const char *ids, const size_t *id_off, const RowSource &src, bool *began) {
  Ctx &c = ix->ctx;
  const size_t d = (size_t)ix->dim;
  const uint32_t ld = ix->ld;
  const uint32_t n_before = ix->n;
  const size_t unranked_before = ix->unranked;
  const bool trace = vt::env::on(vt::env::TRACE_INGEST);
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  // (1) the check.  Fused (the default): the threads that fill the pinned quarters look at every row they copy -- the
  // caller's matrix is read from host memory ONCE (a checker of its own read the same 30 GB beside them, and both slowed
  // down together: 0.85-1.04 s for the rows of 10 M x 768) -- and publish how far the batch is known to be finite
  // (`verified`); the id thread follows that mark and TAKES BACK what it inserted if a later row fails (rollback below:
  // flat.rs:69-85 stores nothing of a rejected batch).  (r04's first form -- a checker on 32 threads beside the copy, the ids
  // only once it had passed -- left the library in r06.)
  const bool fused = src.unvalidated;
  std::atomic<int> checked{src.unvalidated ? 0 : 1};  // 0 running, 1 passed, 2 failed
  std::atomic<size_t> verified{src.unvalidated ? (size_t)0 : count};  // rows [0, verified) are known to be finite
  std::atomic<bool> aborted{false};                   // the copy gave up (device error): nobody will finish the check
  std::atomic<size_t> ids_done{0};                    // ids the id thread has placed (the lock-step test hook waits on it)
  std::atomic<bool> id_exited{false};
  // the id thread could not even reserve its room: the copy need not go on.  (Error precedence, ADVICE r5: the check rides on
  // the copy, so a batch that ALSO holds a non-finite row behind the point the copy had reached reports the host's
  // "out of memory", not flat.rs:69-85's "vector contains a non-finite value".  Either way nothing of the batch is stored;
  // the reference, whose id table is a hash map that aborts the VM when it cannot grow, has no such case to order.)
  std::atomic<bool> id_failed_early{false};
  double t_checked = 0.0;
  // (1a) the batch's ids in bytewise order, for the ranking behind the id table: needs nothing but the bytes, so it starts
  // now (ids that arrive in order -- a snapshot rebuild sorts by id, collection.ex:427-433 -- are found out in one pass)
  std::vector<uint32_t> batch_order;
  std::thread sorter([&] {
    auto id_less_at = [&](uint32_t x, uint32_t y) {
      const size_t lx = id_off[x + 1] - id_off[x], ly = id_off[y + 1] - id_off[y];
      const size_t m = std::min(lx, ly);
      const int cmp = m ? std::memcmp(ids + id_off[x], ids + id_off[y], m) : 0;
      if (cmp) return cmp < 0;
      return lx < ly;
    };
    bool ascending = true;
    for (size_t i = 1; i < count && ascending; ++i) ascending = id_less_at((uint32_t)(i - 1), (uint32_t)i);
    if (ascending) return;
    batch_order.resize(count);
    for (size_t i = 0; i < count; ++i) batch_order[i] = (uint32_t)i;
    parallel_sort(batch_order, id_less_at);
  });
  // (1b) room for the rows.  A slab that is (or starts as) a mapped range grows by mapping 1-GiB chunks behind the rows:
  // 29 of them for 10 M x 768 rows are 0.25 s of hipMemCreate / hipMemMap -- a thread maps them AHEAD of the copy, which
  // only waits when it catches up.  Taken only when the card says it has the room (a growth that fails half way could
  // not be taken back once the ids are in; one that fails up front fails as a whole, below).
  Slab &sl = ix->slab;
  const size_t row_bytes_ = (size_t)ld * sizeof(float);
  const uint64_t rows_up = ((uint64_t)n_before + count + vt::kTileRows - 1) / vt::kTileRows * vt::kTileRows;
  const size_t need_bytes = (size_t)rows_up * row_bytes_;
  std::atomic<size_t> mapped_bytes{0};
  std::atomic<int> map_status{VT_OK};
  std::string map_error;
  std::thread mapper;
  bool progressive = false;
  auto join_helpers = [&]() {
    if (mapper.joinable()) mapper.join();
    if (sorter.joinable()) sorter.join();
  };
  if (Slab::mapping_allowed() && need_bytes >= Slab::chunk_bytes() && need_bytes > sl.bytes && (sl.mapped || (n_before == 0 && sl.p == nullptr))) {
    size_t free_b = 0, total_b = 0;
    const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > (need_bytes - sl.bytes) + 2 * Slab::chunk_bytes();
    (void)hipGetLastError();
    if (room) {
      int st = VT_OK;
      if (!sl.mapped) st = sl.start_mapped(std::min(need_bytes, 2 * Slab::chunk_bytes()), c.device);
      if (st == VT_OK) {
        progressive = true;
        ix->dX = sl.p;
        mapped_bytes.store(sl.bytes);
        mapper = std::thread([&] {
          (void)hipSetDevice(c.device);
          while (sl.bytes < need_bytes) {
            const int ms = no_throw([&]() -> int { return sl.map_up_to(std::min(need_bytes, sl.bytes + sl.chunk), c.device); });
            if (ms != VT_OK) {
              map_error = g_last_error;
              map_status.store(ms);
              return;
            }
            mapped_bytes.store(sl.bytes);
          }
        });
      } else if (st != VT_ERR_UNSUPPORTED) {
        join_helpers();
        return st;
      }
    }
  }
  if (!progressive) {
    const int st = index_reserve(ix, (uint32_t)(n_before + count));
    if (st != VT_OK) {
      join_helpers();
      return st;
    }
    mapped_bytes.store(ix->slab.bytes);
  }
  const double t_room = since();
  // (2) ids and ranks, once the check has passed
  std::vector<uint32_t> target(count);
  bool in_order = true;
  int id_status = VT_OK;
  std::string id_error;
  double t_ids = 0.0, t_ranked = 0.0, t_staged = 0.0;
  bool ranked_here = false, rolled_back = false;
  std::thread idt([&] {
    // (the ids' hashes change nothing in the index: they are formed on a few threads while the check still runs; so is
    // the room in the id tables -- as in the general path no regrowth inside the id loop; reserving it on the calling
    // thread held the first copy back by 40 ms per ten million ids: a 256-MB slot array to clear)
    std::vector<uint64_t> hashes;
    const int hst = no_throw([&]() -> int {
      const size_t need = (size_t)ix->n + count;
      if (need * 10 > ix->row_of.slots() * 7) ix->row_of.reserve(std::max(need, 2 * ix->row_of.size()));
      if (need > ix->ids.capacity()) ix->ids.reserve(std::max(need, 2 * ix->ids.capacity()));
      if (need > ix->rank_host.capacity()) ix->rank_host.reserve(std::max(need, 2 * ix->rank_host.capacity()));
      hashes.resize(count);
      parallel_for(count, 1u << 16, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) hashes[i] = vt_host::hash_id(ids + id_off[i], id_off[i + 1] - id_off[i]);
      }, 8u);
      return VT_OK;
    });
    if (hst != VT_OK) {
      // (no room for the ids: nothing has changed yet -- the copy loop stops at its next quarter, the rows it
      // has already put behind the index are zeroed again below)
      id_status = hst;
      id_error = g_last_error;
      id_failed_early.store(true);
      id_exited.store(true);
      return;
    }
    // what index_row_for changes, for the way back
    const uint32_t s_n = ix->n, s_max_rank = ix->max_rank;
    const uint64_t s_epoch = ix->epoch;
    const bool s_external = ix->external_ranks, s_clean = ix->ranks_clean, s_dirty_all = ix->rank_dirty_all;
    const size_t s_unranked = ix->unranked;
    const std::string s_max_id = ix->max_id;
    std::vector<uint32_t> s_rank_host;
    id_status = no_throw([&]() -> int {
      if (s_external) s_rank_host = ix->rank_host;  // (a new id drops externally supplied ranks: the column is refilled)
      constexpr size_t kAhead = 16;
      size_t done = 0;
      for (;;) {
        const size_t lim = verified.load(std::memory_order_acquire);
        if (done < lim) {
          *began = true;
          for (size_t i = done; i < std::min(lim, done + kAhead); ++i) ix->row_of.prefetch(hashes[i]);
          for (; done < lim; ++done) {
            if (done + kAhead < lim) ix->row_of.prefetch(hashes[done + kAhead]);
            bool is_new = false;
            target[done] = index_row_for(ix, ids + id_off[done], id_off[done + 1] - id_off[done], &is_new, hashes[done]);
            if (!is_new || target[done] != n_before + done) in_order = false;
          }
          ids_done.store(done, std::memory_order_release);
          continue;
        }
        if (done == count) break;
        if (checked.load() == 2 || aborted.load()) {
          // a row behind the mark is not finite (or the copy gave up): the ids that went in come out again, newest first
          if (trace) std::fprintf(stderr, "[vt ingest] the batch is rejected behind row %zu: %u ids taken back\n", done, ix->n - s_n);
          for (uint32_t r = ix->n; r-- > s_n;) {
            const std::string &k = ix->ids[r];
            ix->row_of.erase(k.data(), k.size(), vt_host::hash_id(k.data(), k.size()));
          }
          ix->ids.resize(s_n);
          ix->rank_host.resize(s_n);
          if (s_external) ix->rank_host = s_rank_host;
          ix->n = s_n;
          ix->epoch = s_epoch;
          ix->external_ranks = s_external;
          ix->ranks_clean = s_clean;
          ix->rank_dirty_all = s_dirty_all;
          ix->unranked = s_unranked;
          ix->max_rank = s_max_rank;
          ix->max_id = s_max_id;
          *began = false;
          rolled_back = true;
          return VT_OK;
        }
        std::this_thread::yield();
      }
      t_ids = since();
#ifdef VT_TEST_HOOKS
      if (vt::env::on(vt::env::TEST_FAIL_AFTER_ID_UPDATE)) return fail(VT_ERR_DEVICE, "injected failure after the id table changed");
#endif
      // the host half of the ranking (index_sync_ranks sorts ids; its upload waits for the rows' stream below)
      if (in_order && !ix->ranks_clean && count >= (size_t)n_before / 4) {
        if (sorter.joinable()) sorter.join();
        // (the batch's rows are n_before + i: its sorted ids are the sorted newcomers, if nobody else was waiting for a rank)
        // of them, the rows that are still waiting for a rank -- an ascending run at the start of the batch got ranks as it arrived
        std::vector<uint32_t> fresh;
        if (batch_order.size() == count && unranked_before == 0) {
          // (ten million look-ups all over the rank column: shared out, the pieces put together in order)
          constexpr size_t kParts = 16;
          std::vector<uint32_t> piece[kParts];
          std::vector<std::thread> th;
          for (size_t t = 0; t < kParts; ++t)
            th.emplace_back([&, t] {
              const size_t lo = count * t / kParts, hi = count * (t + 1) / kParts;
              for (size_t i = lo; i < hi; ++i)
                if (ix->rank_host[n_before + batch_order[i]] == kUnranked) piece[t].push_back(n_before + batch_order[i]);
            });
          for (auto &t : th) t.join();
          fresh.reserve(ix->unranked);
          for (size_t t = 0; t < kParts; ++t) fresh.insert(fresh.end(), piece[t].begin(), piece[t].end());
          if (fresh.size() != ix->unranked) fresh.clear();
        }
        VT_TRY(index_sync_ranks(ix, false, /*host_only=*/true, fresh.empty() ? nullptr : &fresh));
        ranked_here = true;
      }
      t_ranked = since();
      return VT_OK;
    });
    if (id_status != VT_OK) id_error = g_last_error;
    id_exited.store(true);
  });
  // (3) the rows, on this thread
  int copy_status = VT_OK;
  {
    const size_t row_bytes = (size_t)ld * sizeof(float);
    // Four pinned quarters of 128 MiB, their DMAs alternating between two streams: two copies are in flight while a
    // third quarter is being filled (one stream = one SDMA queue: 41 GB/s of the link's 64)
    constexpr int kStreams = 2;
    constexpr int kQuarters = 4;
    // (VT_INGEST_STAGE_MB: tests take quarters of 1 MiB, so that a batch of a few MB crosses many of them)
    const size_t kQuarterBytes = [] {  // (read per call: tests set it for one load)
      const long v = vt::env::get(vt::env::INGEST_STAGE_MB);
      return (size_t)(v >= 1 && v <= 1024 ? v : 128) << 20;
    }();
    const size_t stage_rows = std::max<size_t>(1, std::min<size_t>(count, kQuarterBytes / row_bytes));
    copy_status = c.hStage.ensure(kQuarters * stage_rows * row_bytes);
    t_staged = since();
    hipStream_t second = nullptr;
    hipEvent_t done[kQuarters] = {};
    for (int q = 0; q < kQuarters && copy_status == VT_OK; ++q)
      if (hipEventCreateWithFlags(&done[q], hipEventDisableTiming) != hipSuccess) copy_status = fail(VT_ERR_DEVICE, "hipEventCreate (staging)");
    if (copy_status == VT_OK && kStreams == 2 && hipStreamCreateWithFlags(&second, hipStreamNonBlocking) != hipSuccess)
      copy_status = fail(VT_ERR_DEVICE, "hipStreamCreate (staging)");
    bool used[kQuarters] = {};
    size_t i = 0;
    for (int q = 0; i < count && copy_status == VT_OK && checked.load() != 2 && !id_failed_early.load(); q = (q + 1) % kQuarters) {
      float *stage = reinterpret_cast<float *>(c.hStage.p) + (size_t)q * stage_rows * ld;
      const size_t chunk = std::min(stage_rows, count - i);
      if (used[q] && hipEventSynchronize(done[q]) != hipSuccess) copy_status = fail(VT_ERR_DEVICE, "hipEventSynchronize (staging quarter)");
      std::atomic<bool> bad{false};
      parallel_for(chunk, 2048, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
          const float *from = src.host + (i + j) * src.d;
          if (fused && !all_finite_bits(from, d)) {
            bad.store(true);
            return;
          }
          float *dst = stage + j * ld;
          std::memcpy(dst, from, d * sizeof(float));
          for (size_t t = d; t < ld; ++t) dst[t] = 0.0f;
        }
      });
      if (fused) {
        if (bad.load()) {
          t_checked = since();
          checked.store(2);
          break;
        }
        verified.store(i + chunk, std::memory_order_release);
#ifdef VT_TEST_HOOKS
        // (libvettore_hip_hooks.so only: the ids keep step with the verified rows, so a bad row in the last quarter is
        // found with every earlier id in the table -- tests/test_gpu_ingest.py checks that they all come out again)
        if (vt::env::on(vt::env::TEST_INGEST_LOCKSTEP))
          while (ids_done.load(std::memory_order_acquire) < i + chunk && !id_exited.load()) std::this_thread::yield();
#endif
        if (i + chunk == count) {
          t_checked = since();
          checked.store(1);
        }
      }
      // (the chunks behind this quarter are mapped by now, or will be in a moment)
      while (progressive && mapped_bytes.load() < (size_t)(n_before + i + chunk) * row_bytes && map_status.load() == VT_OK) std::this_thread::yield();
      if (map_status.load() != VT_OK) copy_status = fail(map_status.load(), map_error);
      hipStream_t lane = (second && (q & 1)) ? second : c.stream;
      if (copy_status == VT_OK && hipMemcpyAsync(ix->dX + (size_t)(n_before + i) * ld, stage, chunk * row_bytes, hipMemcpyHostToDevice, lane) != hipSuccess)
        copy_status = fail(VT_ERR_DEVICE, "hipMemcpyAsync (rows to the device)");
      if (copy_status == VT_OK && hipEventRecord(done[q], lane) != hipSuccess) copy_status = fail(VT_ERR_DEVICE, "hipEventRecord");
      used[q] = true;
      i += chunk;
    }
    if (second) {
      if (hipStreamSynchronize(second) != hipSuccess && copy_status == VT_OK) copy_status = fail(VT_ERR_DEVICE, "hipStreamSynchronize (staging)");
      (void)hipStreamDestroy(second);
    }
    for (int q = 0; q < kQuarters; ++q)
      if (done[q]) (void)hipEventDestroy(done[q]);
    if (copy_status != VT_OK) aborted.store(true);  // (the id thread must not wait for rows nobody will look at)
    if (copy_status != VT_OK) (void)hipGetLastError();
  }
  const double t_copied = since();
  const std::string copy_error = g_last_error;
  idt.join();  // (first: it joins the sorter itself when it needs the order)
  join_helpers();
  (void)hipStreamSynchronize(c.stream);
  if (progressive) {
    // what index_reserve does after a growth: the slab's new space beyond the rows is zeros, the capacity follows
    const size_t rows_end = std::min(sl.bytes, (size_t)(n_before + count) * row_bytes_);
    if (sl.bytes > rows_end) (void)hipMemsetAsync(reinterpret_cast<char *>(sl.p) + rows_end, 0, sl.bytes - rows_end, c.stream);
    (void)hipStreamSynchronize(c.stream);
    sl.defined = sl.bytes;
    ix->cap = (uint32_t)std::min<uint64_t>(sl.bytes / row_bytes_ / vt::kTileRows * vt::kTileRows, 0xFFFFFFE0ull);
    if (map_status.load() != VT_OK && copy_status == VT_OK) copy_status = fail(map_status.load(), map_error);
  }
  auto zero_free_rows = [&]() {  // what reached the slab behind the index goes again: rows n .. cap are zeros
    const size_t from = (size_t)ix->n * ld, to = (size_t)(n_before + count) * ld;
    if (to > from) {
      (void)hipMemsetAsync(ix->dX + from, 0, (to - from) * sizeof(float), c.stream);
      (void)hipStreamSynchronize(c.stream);
    }
  };
  if (checked.load() == 2) {  // flat.rs:69-85: nothing has been stored
    zero_free_rows();
    if (ix->n == 0) ix->dim = -1;  // (the dimension was set for this batch: a rejected first batch leaves none behind)
    return VT_ERR_NON_FINITE;
  }
  if (id_status != VT_OK) {
    // (*began set: ids went in and something failed behind them -- the caller poisons the handle; not set: the id
    // thread never started on the table, the index is what it was once the free rows are zeros again)
    if (!*began) {
      zero_free_rows();
      if (ix->n == 0) ix->dim = -1;
    }
    return fail(id_status, id_error);
  }
  if (copy_status != VT_OK) {
    if (rolled_back) zero_free_rows();  // (the ids came out again: nothing of the batch stays)
    return fail(copy_status, copy_error);
  }
  if (!in_order) {
    // upserts / duplicates in the batch: the ids are in the table already (target[] says where every row belongs);
    // the general path places the rows -- it finds every id present and changes nothing else
    zero_free_rows();
    return kRetryGeneral;
  }
  // derived columns: everything behind n_before is new
  if (count > kMaxDerivedDirty) {
    ix->bits_valid = false;
    ix->nz_valid = false;
    ix->max_sqnorm = -1.0;
    ix->bits_dirty.clear();
    ix->nz_dirty.clear();
    ix->norm_dirty.clear();
    ix->sh_valid = false;
    ix->sh_dirty.clear();
  } else {
    for (size_t i = 0; i < count; ++i) index_touch_row(ix, target[i]);
  }
  // the rank column on the device
  if (ranked_here || ix->ranks_clean) {
    // (ids that arrived in order extend the ranks in place: only the new ones travel, unless the column has to be regrown)
    uint32_t from = ranked_here ? 0 : n_before;
    if (ix->dRank.count < std::max<size_t>(ix->cap, ix->n)) {
      VT_TRY(ix->dRank.ensure(std::max<size_t>(ix->cap, ix->n)));
      from = 0;
    }
    VT_HIP(hipMemcpyAsync(ix->dRank.p + from, ix->rank_host.data() + from, (size_t)(ix->n - from) * sizeof(uint32_t),
                          hipMemcpyHostToDevice, c.stream));
    VT_HIP(hipStreamSynchronize(c.stream));
    ix->rank_dirty.clear();
    ix->rank_dirty_all = false;
  } else {
    ix->rank_dirty_all = true;  // a small part of a large index: the next search that needs them ranks (as in the general path)
  }
  if (trace)
    std::fprintf(stderr, "[vt ingest] %zu rows, phases overlapped: room %s at %.3f s, pinned staging ready at %.3f s, finiteness check done at %.3f s, id table at %.3f s, id ranks at %.3f s, rows on the device at %.3f s, all at %.3f s\n",
                 count, progressive ? "being mapped ahead of the copy from" : "made by", t_room, t_staged, t_checked, t_ids, t_ranked, t_copied, since());
  return VT_OK;
}

// Shared body of insert / insert_many / load_matrix: rows are already validated.
// `*began` is set once the index has started to change: a failure after that point
// leaves it inconsistent (the caller poisons the handle).
int index_store_rows(Shard *ix, size_t count, const char *ids, const size_t *id_off, const RowSource &src, bool *began) {
  if (count == 0) return VT_OK;
  Ctx &c = ix->ctx;
  const size_t d = (size_t)ix->dim;
  if ((uint64_t)ix->n + count > 0xFFFFFFF0ull) return fail(VT_ERR_UNSUPPORTED, "more than 2^32-16 rows");
  // rows when the call began.  (A bulk load that hands over to this path -- kRetryGeneral: its ids were not all new and
  // distinct -- has already put every id into the table and grown n and rank_host: the rank bookkeeping at the end must
  // cover the rows IT added, not only the ones added below -- none.  ADVICE r4: ids ascending above max_id with one
  // duplicate, or a sorted snapshot reloaded over existing ids plus new ones, kept ranks_clean and left the device
  // column without the new rows' ranks, null on a first load.)
  const uint32_t n_entry = ix->n;
  if (src.host && !src.device && !src.off && !src.pick && count >= 65536) {
    const int st = index_store_bulk_host(ix, count, ids, id_off, src, began);  // (makes room itself, beside the check)
    if (st != kRetryGeneral) return st;
  } else {
    if (src.unvalidated) VT_TRY(validate_matrix(src.host, count, src.d, ix->dim));
    VT_TRY(index_reserve(ix, ix->n + (uint32_t)count));
  }
  const uint32_t n_before = ix->n;
  std::vector<uint32_t> target(count);
  bool all_appended_in_order = true;
  {
    // Room for every new id BEFORE the index starts to change: no rehash / regrowth inside the id
    // loop, and no allocation that could fail between the table's change and the rows' arrival
    // (all sizes; geometric, or a corpus that arrives in many appends re-hashes and re-copies its
    // whole id table at every one of them -- 84 M ids: 15 s).
    const size_t need = (size_t)ix->n + count;
    if (need * 10 > ix->row_of.slots() * 7) ix->row_of.reserve(std::max(need, 2 * ix->row_of.size()));
    if (need > ix->ids.capacity()) ix->ids.reserve(std::max(need, 2 * ix->ids.capacity()));
    if (need > ix->rank_host.capacity()) ix->rank_host.reserve(std::max(need, 2 * ix->rank_host.capacity()));
  }
  *began = true;
  const bool trace = (count > 100000 && vt::env::on(vt::env::TRACE_INGEST)) || vt::env::get(vt::env::TRACE_INGEST) == 2;  // phase timings on stderr (tools/ingest_probe.py)
  const auto t_ids = std::chrono::steady_clock::now();
  {
    // the table's slots are fetched a few ids ahead of their use: a bulk load walks a table far
    // larger than the caches, one miss per id
    constexpr size_t kAhead = 8;
    uint64_t ring[kAhead];
    for (size_t i = 0; i < std::min(count, kAhead); ++i) {
      ring[i] = vt_host::hash_id(ids + id_off[i], id_off[i + 1] - id_off[i]);
      ix->row_of.prefetch(ring[i]);
    }
    for (size_t i = 0; i < count; ++i) {
      const uint64_t hash = ring[i % kAhead];
      if (i + kAhead < count) {
        ring[i % kAhead] = vt_host::hash_id(ids + id_off[i + kAhead], id_off[i + kAhead + 1] - id_off[i + kAhead]);
        ix->row_of.prefetch(ring[i % kAhead]);
      }
      bool is_new = false;
      target[i] = index_row_for(ix, ids + id_off[i], id_off[i + 1] - id_off[i], &is_new, hash);
      if (!is_new || target[i] != n_before + i) all_appended_in_order = false;
    }
  }
  const auto t_rows = std::chrono::steady_clock::now();
#ifdef VT_TEST_HOOKS
  // (test hook, libvettore_hip_hooks.so only: a device failure between the id table's change and
  // the rows' arrival, the one window in which a mutation cannot be taken back --
  // tests/test_gpu_multishard.py checks that the handle is poisoned from then on)
  if (vt::env::on(vt::env::TEST_FAIL_AFTER_ID_UPDATE)) return fail(VT_ERR_DEVICE, "injected failure after the id table changed");
#endif
  if (count > kMaxDerivedDirty) {
    ix->bits_valid = false;
    ix->nz_valid = false;
    ix->max_sqnorm = -1.0;
    ix->bits_dirty.clear();
    ix->nz_dirty.clear();
    ix->norm_dirty.clear();
    ix->sh_valid = false;
    ix->sh_dirty.clear();
  } else {
    for (size_t i = 0; i < count; ++i) index_touch_row(ix, target[i]);
  }
  const uint32_t ld = ix->ld;
  bool pending = false;
  unsigned char *land_buf = nullptr;  // a trickle's slot of the landing ring (host and device view), the event behind it,
  const unsigned char *land_dev = nullptr;  // and the id ranks that land with the rows
  hipEvent_t land_ev = nullptr;
  uint32_t land_rank_first = 0, land_ranks = 0;
  if (src.device) {
    bool picks_dense = true;  // the batch is one contiguous block of the source
    if (src.pick)
      for (size_t i = 1; i < count && picks_dense; ++i) picks_dense = src.pick[i] == src.pick[0] + i;
    const float *first = src.device + (src.pick ? (size_t)src.pick[0] * d : 0);
    // (test hook: a one-GPU box has no other device to own the rows)
    bool foreign = device_of_pointer(src.device) != c.device;
#ifdef VT_TEST_HOOKS
    foreign = foreign || vt::env::on(vt::env::TEST_FOREIGN_ROWS);  // (libvettore_hip_hooks.so only)
#endif
    if (foreign) {
      // Rows that live on another device of the node.  A mapped slab admits no peer copy at all
      // (only this device has been given access to its chunks, hipMemSetAccess), and a copy per
      // row is two API calls and -- through a staging block -- a stream sync per row once the
      // picks are not consecutive, which is the normal case: one matrix dealt to S shards by the
      // hash of its ids leaves every shard every S-th row or so (100x slower than
      // necessary).  So: ONE peer copy per block brings the whole source span pick[i] .. pick[e-1]
      // (<= 256 MB; the rows of the other shards in between ride along) into an ordinary buffer
      // here, one gather launch places this shard's rows from it (zero padded to ld; of an id that
      // appears twice in the batch only the LAST occurrence, flat.rs:270-281), one sync per block.
      const size_t block_rows = std::max<size_t>(1, ((size_t)256 << 20) / (d * sizeof(float)));
      std::unordered_map<uint32_t, size_t> last;
      for (size_t i = 0; i < count; ++i) last[target[i]] = i;
      DevBuf<float> stage;
      DevBuf<uint32_t> dMap;
      std::vector<uint32_t> map;
      size_t i = 0;
      while (i < count) {
        const size_t p0 = src.pick ? src.pick[i] : i;
        size_t e = i + 1, p_hi = p0;
        while (e < count) {  // picks ascending and within one block of the first
          const size_t pe = src.pick ? src.pick[e] : e;
          if (pe < p_hi || pe - p0 >= block_rows) break;
          p_hi = pe;
          ++e;
        }
        const size_t span = p_hi - p0 + 1;
        VT_TRY(stage.ensure(span * d));
        VT_HIP(hipMemcpyAsync(stage.p, src.device + p0 * d, span * d * sizeof(float), hipMemcpyDefault, c.stream));
        map.clear();
        for (size_t j = i; j < e; ++j) {
          if (last[target[j]] != j) continue;
          map.push_back((uint32_t)((src.pick ? src.pick[j] : j) - p0));
          map.push_back(target[j]);
        }
        if (!map.empty()) {
          VT_TRY(dMap.ensure(map.size()));
          VT_HIP(hipMemcpyAsync(dMap.p, map.data(), map.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
          VT_HIP(vt::launch_gather_rows(stage.p, (uint32_t)d, dMap.p, (uint32_t)(map.size() / 2), ix->dX, ld, c.stream));
        }
        VT_HIP(hipStreamSynchronize(c.stream));  // the block and the map are reused
        i = e;
      }
    } else if (all_appended_in_order && picks_dense) {
      // (hipMemcpyDefault: the source may live on another device of the node)
      float *dst = ix->dX + (size_t)n_before * ld;
      if (ld == d) VT_HIP(hipMemcpyAsync(dst, first, count * d * sizeof(float), hipMemcpyDefault, c.stream));
      else VT_HIP(vt::launch_pad_rows(first, (uint32_t)count, (uint32_t)d, dst, ld, c.stream));
    } else if (count < 64) {
      // (few rows: plain copies)
      for (size_t i = 0; i < count; ++i) {
        float *dst = ix->dX + (size_t)target[i] * ld;
        const size_t p = src.pick ? src.pick[i] : i;
        VT_HIP(hipMemsetAsync(dst, 0, (size_t)ld * sizeof(float), c.stream));
        VT_HIP(hipMemcpyAsync(dst, src.device + p * d, d * sizeof(float), hipMemcpyDefault, c.stream));
      }
    } else {
      // scattered rows (upserts, or a batch dealt to shards by the hash of its ids): one gather
      // launch over a (source row, slab row) map instead of a copy per row.  Duplicate ids of a
      // batch map to one slab row: only the LAST occurrence is kept in the map (flat.rs:270-281).
      std::vector<uint32_t> map;
      map.reserve(2 * count);
      std::unordered_map<uint32_t, size_t> last;
      for (size_t i = 0; i < count; ++i) last[target[i]] = i;
      for (size_t i = 0; i < count; ++i) {
        if (last[target[i]] != i) continue;
        map.push_back((uint32_t)(src.pick ? src.pick[i] : i));
        map.push_back(target[i]);
      }
      DevBuf<uint32_t> dMap;
      VT_TRY(dMap.ensure(map.size()));
      VT_HIP(hipMemcpyAsync(dMap.p, map.data(), map.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
      VT_HIP(vt::launch_gather_rows(src.device, (uint32_t)d, dMap.p, (uint32_t)(map.size() / 2), ix->dX, ld, c.stream));
      VT_HIP(hipStreamSynchronize(c.stream));  // (map and dMap die with this scope)
    }
    VT_HIP(hipStreamSynchronize(c.stream));
  } else if ((size_t)count * ((size_t)ld * sizeof(float) + 2 * sizeof(uint32_t)) <= Shard::Landing::kSlotBytes) {
    // A trickle (Vettore.put/2 record by record; up to 21 rows of 768 floats): the rows are staged in a slot of the landing
    // ring -- rows, then the slab row of each, then (below) the id ranks that are to go up with them -- ONE kernel moves
    // all of it (launch_land_rows, queued at the end of this function), and the call returns without waiting for it
    // (Shard::Landing, host/vt_types.h): readers on other streams wait for the slot's event on the device.
    VT_TRY(landing_slot(ix, &land_buf, &land_ev, &land_dev));
    float *stage = reinterpret_cast<float *>(land_buf);
    uint32_t *slab_rows = reinterpret_cast<uint32_t *>(stage + count * ld);
    for (size_t j = 0; j < count; ++j) {
      const size_t p = src.pick ? src.pick[j] : j;
      const float *row = src.off ? src.host + src.off[p] : src.host + p * src.d;
      float *dst = stage + j * ld;
      std::memcpy(dst, row, d * sizeof(float));
      for (size_t t = d; t < ld; ++t) dst[t] = 0.0f;
      slab_rows[j] = target[j];
      // (an id that comes twice in one batch: the LAST occurrence is the row, flat.rs:270-281 -- the blocks of one launch
      // are not ordered, so the earlier ones are marked and skipped)
      for (size_t e = 0; e < j; ++e)
        if (slab_rows[e] == target[j]) slab_rows[e] = 0xFFFFFFFFu;
    }
    pending = true;
  } else {
    // two pinned staging halves: host threads fill one (rows padded to ld) while the DMA
    // of the other is in flight; runs of consecutive target rows go in one copy
    const size_t row_bytes = (size_t)ld * sizeof(float);
    const size_t stage_rows = std::max<size_t>(1, std::min<size_t>(count, (256u << 20) / row_bytes));
    VT_TRY(c.hStage.ensure(2 * stage_rows * row_bytes));
    hipEvent_t done[2] = {c.ev2, c.ev3};
    bool used[2] = {false, false};
    size_t i = 0;
    for (int half = 0; i < count; half ^= 1) {
      float *stage = reinterpret_cast<float *>(c.hStage.p) + (size_t)half * stage_rows * ld;
      const size_t chunk = std::min(stage_rows, count - i);
      if (used[half]) VT_HIP(hipEventSynchronize(done[half]));  // its previous copies have left the buffer
      parallel_for(chunk, 2048, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
          const size_t p = src.pick ? src.pick[i + j] : i + j;
          const float *row = src.off ? src.host + src.off[p] : src.host + p * src.d;
          float *dst = stage + j * ld;
          std::memcpy(dst, row, d * sizeof(float));
          for (size_t t = d; t < ld; ++t) dst[t] = 0.0f;
        }
      });
      size_t j = 0;
      while (j < chunk) {
        size_t e = j + 1;
        while (e < chunk && target[i + e] == target[i + e - 1] + 1) ++e;
        VT_HIP(hipMemcpyAsync(ix->dX + (size_t)target[i + j] * ld, stage + j * ld, (e - j) * row_bytes,
                              hipMemcpyHostToDevice, c.stream));
        j = e;
      }
      VT_HIP(hipEventRecord(done[half], c.stream));
      used[half] = true;
      i += chunk;
    }
    pending = true;  // one wait at the end of the function covers the rows and their ranks
  }
  // A bulk load ranks its ids right away (the load itself takes far longer) -- unless it is a
  // small part of what is already there: re-ranking costs a pass over ALL ids, and a corpus that
  // arrives in many appends would pay it every time (84 M rows in 21 appends: 14 s each); the
  // next search does it once.  Trickling inserts leave their rows unranked for the lazy search path.
  const bool rank_now = !ix->ranks_clean && count >= kBulkRankRows && count >= (size_t)n_entry / 4;
  if (!ix->ranks_clean && !rank_now) {
    // the device column is brought up to date lazily (index_lazy_ranks) or by the next re-rank
    if (count < kBulkRankRows)
      for (uint32_t r = n_entry; r < ix->n; ++r) ix->rank_dirty.push_back(r);
    else
      ix->rank_dirty_all = true;
    if (ix->rank_dirty.size() > kMaxDirtyRanks) ix->rank_dirty_all = true;
  }
  // keep device ranks current when they stayed valid (sorted appends)
  if (ix->ranks_clean && ix->n > n_entry) {
    uint32_t from = n_entry;
    if (ix->dRank.count < std::max<size_t>(ix->cap, ix->n)) {
      VT_TRY(ix->dRank.ensure(std::max<size_t>(ix->cap, ix->n)));
      from = 0;
    }
    if (land_dev && ix->n - from <= count) {
      // a trickle of sorted appends (one rank per call): the ranks ride in the landing slot, behind the rows and their
      // slab rows, and go up in the same launch
      uint32_t *ranks = reinterpret_cast<uint32_t *>(land_buf + (size_t)count * ld * sizeof(float)) + count;
      std::memcpy(ranks, ix->rank_host.data() + from, (size_t)(ix->n - from) * sizeof(uint32_t));
      land_rank_first = from;
      land_ranks = ix->n - from;
    } else if (ix->n - from <= 64) {
      // a few sorted appends: through a pinned block of their own (rank_host is pageable: the runtime would stage the
      // copy itself), waited for below
      land_ev = nullptr;
      VT_TRY(c.hRankStage.ensure(64));
      std::memcpy(c.hRankStage.p, ix->rank_host.data() + from, (size_t)(ix->n - from) * sizeof(uint32_t));
      VT_HIP(hipMemcpyAsync(ix->dRank.p + from, c.hRankStage.p, (size_t)(ix->n - from) * sizeof(uint32_t), hipMemcpyHostToDevice,
                            c.stream));
    } else {
      land_ev = nullptr;  // (a copy from pageable memory: waited for below)
      VT_HIP(hipMemcpyAsync(ix->dRank.p + from, ix->rank_host.data() + from, (size_t)(ix->n - from) * sizeof(uint32_t),
                            hipMemcpyHostToDevice, c.stream));
    }
    pending = true;
  }
  // a trickle: one launch moves its rows (and ranks) out of the slot; it returns with that launch queued and an event
  // behind it.  Everything else waits here, once.
  if (land_dev)
    VT_HIP(vt::launch_land_rows(reinterpret_cast<const float *>(land_dev), (uint32_t)count, ld, ix->dX, land_ranks ? ix->dRank.p : nullptr,
                                land_rank_first, land_ranks, c.stream));
  if (pending && land_ev) VT_TRY(landing_record(ix, land_ev));
  else if (pending) VT_HIP(hipStreamSynchronize(c.stream));
  const auto t_rank = std::chrono::steady_clock::now();
  if (rank_now) VT_TRY(index_sync_ranks(ix, false));
  if (trace) {
    auto s_of = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    if (count < 1000)  // (VT_TRACE_INGEST=2: a trickle, in microseconds)
      std::fprintf(stderr, "[vt ingest] %zu rows: id table %.1f us, rows to the device %.1f us, id ranks %.1f us\n", count,
                   1e6 * s_of(t_ids, t_rows), 1e6 * s_of(t_rows, t_rank), 1e6 * s_of(t_rank, std::chrono::steady_clock::now()));
    else
      std::fprintf(stderr, "[vt ingest] %zu rows: id table %.3f s, rows to the device %.3f s, id ranks %.3f s\n", count,
                   s_of(t_ids, t_rows), s_of(t_rows, t_rank), s_of(t_rank, std::chrono::steady_clock::now()));
  }
  return VT_OK;
}

int make_hits(const Shard *ix, const std::vector<vt::Entry> &entries, vt_hits **out) {
  auto h = std::make_unique<vt_hits>();
  const size_t m = entries.size();
  if (merge_request_of(ix)) {
    h->by_row_of = ix;
    h->rows.reserve(m);
    h->raw.reserve(m);
    h->rank_key.reserve(m);
    for (const auto &e : entries) {
      h->rows.push_back(e.row);
      h->raw.push_back(e.raw);
      h->rank_key.push_back(rank_key_of(e.key));
    }
    *out = h.release();
    return VT_OK;
  }
  if (m >= (1u << 17)) {
    // limits in the hundreds of thousands (flat.ex:98-103 allows them): the id copies are most of
    // the call -- one string per hit, picked from all over the table -- so they go on several threads
    h->ids.resize(m);
    h->raw.resize(m);
    h->rank_key.resize(m);
    parallel_for(m, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) {
        h->ids[i] = ix->ids[entries[i].row];
        h->raw[i] = entries[i].raw;
        h->rank_key[i] = rank_key_of(entries[i].key);
      }
    });
  } else {
    h->ids.reserve(m);
    h->raw.reserve(m);
    h->rank_key.reserve(m);
    for (const auto &e : entries) {
      h->ids.push_back(ix->ids[e.row]);
      h->raw.push_back(e.raw);
      h->rank_key.push_back(rank_key_of(e.key));
    }
  }
  *out = h.release();
  return VT_OK;
}

int empty_hits(vt_hits **out) {
  *out = new vt_hits();
  return VT_OK;
}

}  // namespace
