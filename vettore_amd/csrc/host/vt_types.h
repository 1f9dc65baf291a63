// vt_types.h -- per-reader context (Ctx), a shard, its worker thread, the RCCL entry points and the handle (vt_flat).
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// Per-device execution context: stream, scratch, profiling.
struct Ctx {
  int device = 0;
  int num_cus = 256;
  int blocks_per_cu = 2;
  int hamming_blocks_per_cu = 2;
  hipStream_t stream = nullptr;
  DevBuf<float> dQ;
  PinnedBuf<uint32_t> hRankStage;  // the ranks of a few appended rows on their way to the device column (index_store_rows)
  // where the kernels of the current call read the query from (upload_query): c.dQ.p after a copy, or -- under
  // VT_DIRECT_QUERY=1, an A/B that measured no gain -- the pinned staging block itself through its host mapping
  const float *qsrc = nullptr;
  uint64_t *dQbits = nullptr;  // the query's sign bits, behind the query in qsrc (upload_query with_bits)
  int qbits_kind = 0;          // what dQbits holds for the query last uploaded: 0 nothing, 1 sign bits, 2 non-zero bits
  DevBuf<uint64_t> dPartKeys;
  DevBuf<vt::Payload> dPartPay;
  DevBuf<uint64_t> dSelKeys;  // second level of the two-level select: kSelGroups * kMaxFusedK entries
  DevBuf<vt::Payload> dSelPay;
  DevBuf<int> dStatus;  // "metric overflow" flag: set by scan kernels, moved out and cleared by the select kernel
  DevBuf<int> dFlag;    // scratch flag of the ingest kernels
  DevBuf<ResultBlock> dStage;  // stage-1 winners of quantized_search, consumed on the device
  // K4h: distance column, two alternating histograms, list counter
  DevBuf<uint16_t> dDist16;
  DevBuf<uint64_t> dListKeys;  // candidate set of > 256 rows kept on the device
  DevBuf<uint32_t> dRankPairs;  // (row, rank) updates of the lazy rank path
  PinnedBuf<uint32_t> hRankPairs;
  DevBuf<uint64_t> dKeyCol;    // one key per row (limits above kMaxFusedK: radix threshold instead of wave buffers)
  DevBuf<vt::Payload> dPayCol; // (row, raw) per row beside the key column (limits above kSelListMax)
  PinnedBuf<uint64_t> hListKeys;
  PinnedBuf<vt::Payload> hListPay;
  DevBuf<uint32_t> dRadixHist, dRadixCount;
  DevBuf<vt::Payload> dListPay;
  DevBuf<uint32_t> dHamHist, dHamCount;
  uint32_t ham_parity = 0;
  bool ham_ready = false, ham_dirty = false;
  // batched search (K2)
  DevBuf<float> dBQ, dBTau, dBSample;
  DevBuf<vt::BatchCand> dBCand;
  DevBuf<uint32_t> dBCount, dBOutCount;
  DevBuf<vt::Entry> dBOut;
  DevBuf<unsigned long long> dBNorm;
  DevBuf<unsigned char> dBQimage;  // K2b: the batch's queries in bf16, fragment order
  // grouped quantized searches: one stage-1 block per query
  DevBuf<ResultBlock> dStageB;
  PinnedBuf<float> hBQ, hBTau;
  PinnedBuf<uint32_t> hBCount, hBOutCount;
  PinnedBuf<vt::Entry> hBOut;
  hipEvent_t ev2 = nullptr, ev3 = nullptr;
  DevBuf<uint32_t> dRows;
  DevBuf<uint64_t> dCandKeys;
  DevBuf<vt::Payload> dCandPay;
  PinnedBuf<float> hQ;
  PinnedBuf<ResultBlock> hRes;  // written by the select kernel through the host mapping
  PinnedBuf<ResultBlock> hFirst;  // a staged search's first-stage block, copied out for a cross-shard merge
  // limits above kMaxFusedK: up to kSelListMax sorted entries + header, host-mapped, allocated on first use
  PinnedBuf<unsigned char> hBig;
  unsigned char *dBigMapped = nullptr;
  ResultBlock *dResMapped = nullptr;
  PinnedBuf<uint32_t> hShard;  // shard of each winner of a cross-shard merge (host mapped)
  uint32_t *dShardMapped = nullptr;
  PinnedBuf<unsigned char> hStage;
  uint64_t seen_landing = 0;  // Shard::landing.seq this context's stream has been made to wait for (see_writes)
  uint32_t begin_rows = 0, begin_dim = 0;  // scan of the last vt_flat_search_begin (profiling)
  bool profiling = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_wait = nullptr;  // wait_exchange: behind a shard's all-gather (created on first use)
  double wait_expect_us = 0.0;   // ... and how long the last waits there took (the worker sleeps through most of it)
  vt_profile prof{};

  ~Ctx() {
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev2) (void)hipEventDestroy(ev2);
    if (ev3) (void)hipEventDestroy(ev3);
    if (ev_wait) (void)hipEventDestroy(ev_wait);
    if (stream) (void)hipStreamDestroy(stream);
  }

  int init(int dev) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return fail(VT_ERR_DEVICE, "no HIP device visible: libvettore_hip has no CPU fallback");
    if (dev < 0 || dev >= ndev) return fail(VT_ERR_DEVICE, "device ordinal out of range");
    device = dev;
    VT_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    VT_HIP(hipGetDeviceProperties(&prop, dev));
    num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    VT_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    VT_HIP(hipEventCreate(&ev0));
    VT_HIP(hipEventCreate(&ev1));
    VT_HIP(hipEventCreate(&ev2));
    VT_HIP(hipEventCreate(&ev3));
    VT_TRY(dStatus.ensure(1));
    VT_TRY(dFlag.ensure(1));
    VT_TRY(dSelKeys.ensure((size_t)vt::kSelGroups * vt::kMaxFusedK));
    VT_TRY(dSelPay.ensure((size_t)vt::kSelGroups * vt::kMaxFusedK));
    VT_HIP(hipMemset(dStatus.p, 0, sizeof(int)));
    VT_TRY(hRes.ensure(1));
    VT_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&dResMapped), hRes.p, 0));
    VT_TRY(hShard.ensure(vt::kMaxFusedK));
    VT_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&dShardMapped), hShard.p, 0));
    return VT_OK;
  }
  int bind() {
    VT_HIP(hipSetDevice(device));
    return VT_OK;
  }
  uint32_t resident_waves() const { return (uint32_t)(num_cus * blocks_per_cu * vt::kWavesPerBlock); }
  // a prefix scan timed with ev0/ev1 but not yet read back (the chained funnel waits once, at its end)
  uint32_t prefix_pending = 0;
  int settle_prefix_profile() {
    if (!prefix_pending) return VT_OK;
    float ms = 0.0f;
    VT_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    prof.prefix_launches += 1;
    prof.prefix_ms += ms;
    prefix_pending = 0;
    return VT_OK;
  }
  // Tiles are dealt to waves statically, so the grid must be fully resident:
  // blocks per CU = what LDS admits, capped (VT_BLOCKS_PER_CU overrides).
  uint32_t grid_for(uint32_t units, size_t lds_bytes, int max_per_cu = 0) const {
    size_t per_cu = lds_bytes ? (160 * 1024) / lds_bytes : 8;
    per_cu = std::max<size_t>(1, std::min<size_t>(per_cu, (size_t)(max_per_cu > 0 ? max_per_cu : blocks_per_cu)));
    const uint32_t want = (units + vt::kWavesPerBlock - 1) / vt::kWavesPerBlock;
    return std::max<uint32_t>(1, std::min<uint32_t>(want, (uint32_t)(num_cus * per_cu)));
  }
};

}  // namespace

struct Shard;
struct vt_hits {
  std::vector<std::string> ids;
  std::vector<float> raw;
  std::vector<uint32_t> rank_key;
  // A shard's contribution to a cross-shard merge (batch_multi) names its rows instead of copying their ids: the
  // merge compares the id bytes where they live and copies only the winners' (r05).  Never handed to a caller.
  const Shard *by_row_of = nullptr;
  std::vector<uint32_t> rows;
};

// What a shard's worker asks of the batch it is about to run for a sharded handle's cross-shard merge (batch_multi): hit
// lists that name their rows instead of copying ids, and one flag per query, set -- with release order -- once that query's
// list is final, so that the handle's calling thread can merge the shards' lists of such queries while later groups still
// run.  Per CALL and per THREAD (r06, ADVICE r5: these were two mutable fields of the shared Shard): the request lives on
// the worker's stack for the length of its job, and a reader on any other thread -- a leased context, a search that
// arrives meanwhile -- never sees it, so a by-row list cannot reach a caller.
struct MergeRequest {
  const Shard *shard;
  std::atomic<unsigned char> *final_flags;  // [queries of the batch]
};
static thread_local const MergeRequest *t_merge_request = nullptr;
struct MergeRequestScope {
  MergeRequest req;
  MergeRequestScope(const Shard *s, std::atomic<unsigned char> *flags) : req{s, flags} { t_merge_request = &req; }
  ~MergeRequestScope() { t_merge_request = nullptr; }
  MergeRequestScope(const MergeRequestScope &) = delete;
  MergeRequestScope &operator=(const MergeRequestScope &) = delete;
};
inline const MergeRequest *merge_request_of(const Shard *ix) {
  return t_merge_request && t_merge_request->shard == ix ? t_merge_request : nullptr;
}

// One shard = one GPU's share of the rows: the slab, its derived columns, the ids of
// its rows.  A plain index has exactly one; vt_flat_new_sharded deals rows to several.
struct Shard {
  Ctx ctx;  // primary context: mutations and derived-data upkeep run here, under the exclusive lock
  // Further contexts (own stream, scratch, result block) so that several readers can be
  // in flight on one handle (the reference's RwLock readers, nifs.rs:304-308); created
  // on demand, handed out by CtxLease.
  std::mutex pool_mu;
  std::condition_variable pool_cv;
  std::vector<std::unique_ptr<Ctx>> extra;
  std::vector<Ctx *> free_ctx;
  bool ctx0_busy = false;
  int metric = 0;
  int order = default_order();
  int nominate = default_nominate();  // VT_NOMINATE_*
  // corpus
  uint32_t n = 0, cap = 0;
  long dim = -1;    // FlatIndex.dimension (None = -1)
  uint32_t ld = 0;  // row stride of the slab in floats = padded_dim(dim), multiple of 64
  Slab slab;
  float *dX = nullptr;  // == slab.p
  DevBuf<uint32_t> dRank;
  DevBuf<uint64_t> dBits;
  bool bits_valid = false;
  // Rows mutated since the bit matrix / the norms were last brought up to date; patched in
  // place at the next use (a full rebuild is a pass over the whole corpus).
  std::vector<uint32_t> bits_dirty, norm_dirty;
  // The same for float hamming / jaccard collections: one bit per coordinate, set iff it is
  // non-zero -- all those two metrics look at (distances.rs:319-347) -- in K4's layout; flat_search
  // reads these 1/32 of the row bytes instead of the rows (vt_search.h, pattern_search_applies).
  DevBuf<uint64_t> dNzBits;
  bool nz_valid = false;
  bool nz_refused = false;  // the card had no room for the column: searches keep reading the rows (until the index is emptied)
  std::vector<uint32_t> nz_dirty;
  // K2s: the rows once more, rounded to bf16, in the order the matrix cores take their operands in
  // (vt_device.h shadow_index) -- what the bf16 nomination pass reads instead of the f32 rows when
  // the card has room for it (vt_search.h, index_ensure_shadow).  Kept like the bit columns: built
  // by the first batch that wants it, patched per mutated row, given back when the slab needs the room.
  DevBuf<uint16_t> dShadow;
  int shadow_mode = default_shadow();  // VT_SHADOW_*
  // (atomic: search_direct looks at it before it takes the handle's lock, vt_flat_set_single_nominate writes it under the exclusive one)
  std::atomic<int> single_nominate{default_single_nominate()};  // vt_flat_set_single_nominate: lone searches through the shadow
  bool sh_valid = false;
  bool sh_refused = false;  // no room (or the slab took the room back): batches stream the f32 rows until the index is emptied
  std::vector<uint32_t> sh_dirty;
  double max_sqnorm = -1.0;  // max_i sum_j x_ij^2, < 0 = stale (error margin of the batched path)
  DevBuf<float> dXnorm2;     // per-row squared norms, valid with max_sqnorm
  // ids
  std::vector<std::string> ids;  // by row
  vt_host::IdTable row_of{&ids};  // id bytes -> row (the bytes themselves stay in `ids`)
  std::vector<uint32_t> rank_host;  // by row
  bool ranks_clean = true;          // rank_host/dRank describe the current rows
  // While !ranks_clean: rows whose device rank differs from rank_host (newcomers carry
  // kUnranked, a swap-delete moved a rank); rank_dirty_all = re-upload the whole column.
  std::vector<uint32_t> rank_dirty;
  bool rank_dirty_all = false;
  size_t unranked = 0;  // rows carrying kUnranked: past a bound the next search rebuilds instead of going lazy
  bool external_ranks = false;      // rank column supplied by vt_flat_set_id_ranks (valid until the next mutation)
  uint64_t epoch = 0;               // bumped by every mutation of the row set (insert of a new id, delete)
  uint64_t external_epoch = 0;      // epoch at which the external ranks were installed
  bool external_expected = false;   // vt_flat_set_id_ranks has been used on this shard: search_begin insists on current ranks
  std::string max_id;               // upper bound of all ids while ranks_clean
  uint32_t max_rank = 0;

  // Trickle mutations land BEHIND the call (r06; the reference's flat_insert is a hash-map insert, nifs.rs:259-271).  A
  // one-row insert, upsert or delete used to end in a wait for its copies under the exclusive lock: ~12 of its 13-25 us.
  // Now the row (and its rank) is staged in one of eight pinned slots, the copies are queued on the primary context's
  // stream, an event is recorded behind them and the call returns.  Whoever reads next on another stream -- a leased
  // reader context, a second context of a grouped path -- makes that stream wait for the newest event first
  // (see_writes: one hipStreamWaitEvent per context and mutation, no host wait; the primary stream is in order by
  // itself).  `seq` and `newest` change under the handle's exclusive lock and are read under its shared lock.
  struct Landing {
    static constexpr int kSlots = 8;
    static constexpr size_t kSlotBytes = 64u << 10;
    PinnedBuf<unsigned char> stage;  // kSlots * kSlotBytes, on first use
    hipEvent_t ev[kSlots] = {};
    bool used[kSlots] = {};
    int next = 0;
    uint64_t seq = 0;
    hipEvent_t newest = nullptr;
  } landing;

  ~Shard() {
    (void)hipSetDevice(ctx.device);
    if (landing.newest) (void)hipStreamSynchronize(ctx.stream);  // (copies out of the pinned slots may still be queued)
    for (hipEvent_t e : landing.ev)
      if (e) (void)hipEventDestroy(e);
    slab.release();
  }

  template <class F>
  void for_each_ctx(F f) {
    f(ctx);
    for (auto &e : extra) f(*e);
  }
};

namespace {

constexpr size_t kMaxContexts = 8;  // readers in flight per shard

// A context for one reader: the primary one if free, else a spare, else a new one (up to
// kMaxContexts), else wait (vt_host::LeaseT, host/vt_concurrency.h).  Held only under the handle's
// shared lock.
struct CtxLease : vt_host::LeaseT<Shard, Ctx> {
  explicit CtxLease(Shard *s)
      : vt_host::LeaseT<Shard, Ctx>(s, kMaxContexts, [](Shard *ix, int *status) -> std::unique_ptr<Ctx> {
          auto nc = std::make_unique<Ctx>();
          *status = nc->init(ix->ctx.device);
          if (*status != VT_OK) return nullptr;
          nc->profiling = ix->ctx.profiling;
          return nc;
        }) {}
};

inline std::unique_ptr<Ctx> make_reader_ctx(Shard *ix, int *status) {
  auto nc = std::make_unique<Ctx>();
  *status = nc->init(ix->ctx.device);
  if (*status != VT_OK) return nullptr;
  nc->profiling = ix->ctx.profiling;
  return nc;
}
// A second context beside the one a reader holds (vt_host::SpareLeaseT): null when none is free.
struct SpareCtxLease : vt_host::SpareLeaseT<Shard, Ctx> {
  explicit SpareCtxLease(Shard *s) : vt_host::SpareLeaseT<Shard, Ctx>(s, kMaxContexts, make_reader_ctx) {}
};

// A slot of the landing ring (Shard::Landing): its buffer once whatever was copied out of it last time has left, and the
// event to record behind the copies queued from it.
int landing_slot(Shard *ix, unsigned char **buf, hipEvent_t *ev, const unsigned char **buf_dev = nullptr) {
  Shard::Landing &l = ix->landing;
  if (!l.stage.p) {
    VT_TRY(l.stage.ensure((size_t)Shard::Landing::kSlots * Shard::Landing::kSlotBytes));
    for (hipEvent_t &e : l.ev) VT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const int s = l.next;
  l.next = (s + 1) % Shard::Landing::kSlots;
  if (l.used[s]) VT_HIP(hipEventSynchronize(l.ev[s]));  // (eight mutations ago: long done)
  l.used[s] = true;
  *buf = l.stage.p + (size_t)s * Shard::Landing::kSlotBytes;
  *ev = l.ev[s];
  if (buf_dev) {  // (the slot as a kernel sees it: the landing kernel reads the rows straight out of the pinned block)
    unsigned char *m = l.stage.mapped();
    if (!m) return fail(VT_ERR_DEVICE, "hipHostGetDevicePointer (landing ring)");
    *buf_dev = m + (size_t)s * Shard::Landing::kSlotBytes;
  }
  return VT_OK;
}
// ... recorded: the mutation's device work is queued on the primary stream, nothing is waited for.
int landing_record(Shard *ix, hipEvent_t ev) {
  VT_HIP(hipEventRecord(ev, ix->ctx.stream));
  ix->landing.newest = ev;
  ix->landing.seq += 1;
  return VT_OK;
}
// Before a reader's first kernel on a stream other than the primary one: that stream waits (on the device) for the
// newest landed mutation.  Under the shared or the exclusive lock.
int see_writes(Shard *ix, Ctx &c) {
  if (&c == &ix->ctx || c.seen_landing == ix->landing.seq) return VT_OK;
  VT_HIP(hipStreamWaitEvent(c.stream, ix->landing.newest, 0));
  c.seen_landing = ix->landing.seq;
  return VT_OK;
}
// what a reader context does before its first kernel
int reader_ready(Shard *ix, Ctx &c) {
  VT_TRY(c.bind());
  return see_writes(ix, c);
}

// ---- RCCL, loaded on first use (librccl is half a gigabyte: a single-GPU index never maps it)
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
  bool ok = false;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
      r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      r.error = std::string("librccl not loadable: ") + (dlerror() ? dlerror() : "?");
      return;
    }
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.ok = r.CommInitAll && r.CommDestroy && r.CommCount && r.AllGather && r.GetErrorString;
    if (!r.ok) r.error = "librccl lacks an expected symbol";
  });
  return r;
}

// One thread per shard of a multi-shard index, bound to the shard's device (vt_host::WorkerT,
// host/vt_concurrency.h): the caller posts the same job to all of them, so the launch overheads
// of the shards overlap and each shard's kernels are issued by a thread whose current device
// never changes.
struct HipWorkerPolicy {
  static void thread_start(int device) { (void)hipSetDevice(device); }
  static int run(const std::function<int()> &fn, std::string *error) {
    g_last_error.clear();
    const int st = guarded(fn);
    if (st != VT_OK) *error = g_last_error;
    return st;
  }
};
using Worker = vt_host::WorkerT<HipWorkerPolicy>;

}  // namespace

// The handle behind the C ABI: FlatResource(RwLock<FlatIndex>) (flat.rs:13-17, nifs.rs:254-257).
struct vt_flat {
  // searches share, mutations exclude (nifs.rs:266-309)
  mutable std::shared_mutex rw;
  // A mutation that failed on the device after it had begun changing the index leaves it
  // poisoned, like a panic under the reference's write lock: every later call fails with
  // "flat lock poisoned" (nifs.rs:269).
  // (atomic: a reader that finds the exchange wedged sets it under the shared lock)
  std::atomic<bool> poisoned{false};
  // an exchange timed out (wait_exchange): the shard streams still hold the stuck collective, and anything that
  // synchronises with them -- hipStreamDestroy, hipFree -- would block for ever.  vt_flat_free then LEAKS the handle
  // (the process is expected to end).
  std::atomic<bool> wedged{false};
  int metric = 0;
  long dim = -1;  // FlatIndex.dimension across all shards
  std::vector<std::unique_ptr<Shard>> shards;
  // multi-shard only
  std::vector<std::unique_ptr<Worker>> workers;
  std::mutex post_mu;  // jobs reach every worker's queue in one order (collectives must match up)
  int exchange = VT_EXCHANGE_HOST;
  bool exchange_forced = false;
  std::string exchange_note;  // which exchange the handle chose when it was made, and why (vt_flat_exchange_note)
  std::vector<ncclComm_t> comms;
  bool comms_tried = false;
  size_t exch_limit = 0;  // entries the exchange blocks are sized for
  std::vector<void *> dBlock, dGather;  // per shard: its own result block, the gathered blocks of all shards
  PinnedBuf<unsigned char> hGather;     // shard 0's gathered copy, read by the merging host thread
  vt_profile xprof{};                   // exchange timing (merge_launches / merge_ms)

  // Searches that arrive while another one is running wait for it and then go TOGETHER: one
  // sweep of the corpus answers up to eight of them (K1m), the matrix-core pass up to 256 (K2),
  // where the same callers on their own streams would read the whole corpus once each.  See
  // coalesced_search.
  using Waiting = vt_host::Waiting;
  using Coalescer = vt_host::Coalescer;
  Coalescer co;
  // funnel_search callers travel together only with equal (stages, candidates): the shapes seen on
  // this handle, by index -- the index is what a waiter carries as its `aux`
  struct FunnelShape {
    std::vector<size_t> stages;
    size_t candidates;
  };
  std::mutex funnel_mu;
  std::vector<FunnelShape> funnel_shapes;
  std::atomic<uint64_t> approx_bytes{0};  // rows x row stride, refreshed by mutations (the coalescer's only use of it is a size class)
  std::atomic<uint64_t> approx_rows{0};   // rows, likewise

  bool multi() const { return shards.size() > 1 || !workers.empty(); }
  size_t total() const {
    size_t t = 0;
    for (auto &s : shards) t += s->n;
    return t;
  }
  ~vt_flat() {
    workers.clear();  // joins the threads before their shards go away
    for (size_t i = 0; i < comms.size(); ++i)
      if (comms[i]) (void)rccl().CommDestroy(comms[i]);
    for (size_t i = 0; i < dBlock.size(); ++i) {
      (void)hipSetDevice(shards[i]->ctx.device);
      if (dBlock[i]) (void)hipFree(dBlock[i]);
      if (dGather[i]) (void)hipFree(dGather[i]);
    }
  }
};
