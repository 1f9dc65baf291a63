// vt_index.cpp -- host side of libvettore_hip.so: the C ABI of
// include/vettore_flat.h, the id table, and the orchestration of the gfx950
// kernels in vt_device.hip.  It mirrors native/vettore/src/nifs.rs (boundary),
// flat.rs (index semantics) and search.rs (stateless helpers) of the reference;
// all metric arithmetic and all selection run on the GPU.
#include "../../include/vettore_flat.h"
#include "vt_device.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

thread_local std::string g_last_error;
// Lane order of wide::f32x8::reduce_add assumed for new indexes (include/vettore_flat.h;
// DESIGN.md "summation order" says why SSE2 and how to pin it): VT_REDUCE_ORDER overrides.
int initial_order() {
  const char *e = std::getenv("VT_REDUCE_ORDER");
  if (e) {
    const std::string v(e);
    if (v == "pair" || v == "0") return VT_ORDER_PAIR;
    if (v == "avx" || v == "1") return VT_ORDER_AVX;
    if (v == "seq" || v == "2") return VT_ORDER_SEQ;
    if (v == "sse2" || v == "3") return VT_ORDER_SSE2;
  }
  return VT_ORDER_SSE2;
}
int g_default_order = initial_order();

int fail(int status, const std::string &detail) {
  g_last_error = detail;
  return status;
}

// (a failed call also stays behind as the thread's "last error", which the launch wrappers
// read after their <<<>>>: it is cleared here, or the next launch would report it again)
#define VT_HIP(expr)                                                                       \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      (void)hipGetLastError();                                                             \
      return fail(VT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));       \
    }                                                                                      \
  } while (0)

// No exception may cross the C ABI: allocation failures and anything unexpected
// become statuses.
template <typename F>
int guarded(F &&f) noexcept {
  try {
    (void)hipGetLastError();  // whatever another library left behind on this thread is not ours to report
    return f();
  } catch (const std::bad_alloc &) {
    return fail(VT_ERR_NOMEM, "out of host memory");
  } catch (const std::exception &e) {
    return fail(VT_ERR_DEVICE, e.what());
  } catch (...) {
    return fail(VT_ERR_DEVICE, "unknown exception");
  }
}

#define VT_TRY(expr)          \
  do {                        \
    int _s = (expr);          \
    if (_s != VT_OK) return _s; \
  } while (0)

inline uint32_t round_up_u32(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }
constexpr size_t kBulkRankRows = 16384;   // an insert of at least this many rows re-ranks at once
constexpr size_t kMaxDirtyRanks = 16384;  // above this the whole rank column is re-uploaded
constexpr uint32_t kUnranked = 0xFFFFFFFFu;  // id_rank of a row inserted out of id order, until the next re-rank

// flat.rs:136-144 validate_vector: empty, then dimension, then finiteness.
int validate_vector(const float *v, size_t n, long dimension) {
  if (n == 0) return VT_ERR_EMPTY;
  if (dimension >= 0 && n != (size_t)dimension) return VT_ERR_DIMENSION;
  for (size_t i = 0; i < n; ++i)
    if (!std::isfinite(v[i])) return VT_ERR_NON_FINITE;
  return VT_OK;
}

int validate_finite(const float *v, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!std::isfinite(v[i])) return VT_ERR_NON_FINITE;
  return VT_OK;
}

// Splits [0, n) over up to 16 host threads (bulk ingest: validation and staging copies
// are plain memory passes).  `f(lo, hi)` must not throw.
template <class F>
void parallel_for(size_t n, size_t grain, F f) {
  unsigned threads = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  if (grain == 0) grain = 1;
  threads = (unsigned)std::min<size_t>(threads, n / grain);
  if (threads <= 1) {
    f((size_t)0, n);
    return;
  }
  const size_t per = (n + threads - 1) / threads;
  std::vector<std::thread> pool;
  pool.reserve(threads);
  for (unsigned t = 0; t < threads; ++t) {
    const size_t lo = (size_t)t * per, hi = std::min(n, lo + per);
    if (lo < hi) pool.emplace_back([lo, hi, &f] { f(lo, hi); });
  }
  for (auto &th : pool) th.join();
}

// validate_vector over the rows of a dense matrix; the error of the FIRST failing row
// (flat.rs:69-85 checks the batch in order).
int validate_matrix(const float *rows, size_t count, size_t d, long expected) {
  if (count == 0) return VT_OK;
  if (d == 0) return VT_ERR_EMPTY;
  if (expected >= 0 && d != (size_t)expected) return VT_ERR_DIMENSION;
  std::mutex mu;
  size_t first_bad = count;
  parallel_for(count, 4096, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      if (validate_finite(rows + i * d, d) != VT_OK) {
        std::lock_guard<std::mutex> g(mu);
        first_bad = std::min(first_bad, i);
        return;
      }
    }
  });
  return first_bad < count ? VT_ERR_NON_FINITE : VT_OK;
}

inline bool id_less(const std::string &a, const std::string &b) { return a < b; }  // bytewise, like Rust String::cmp

// Sorts `idx` with `less` on several threads (chunk sort + pairwise merges).
template <typename Less>
void parallel_sort(std::vector<uint32_t> &idx, Less less) {
  const size_t n = idx.size();
  unsigned hw = std::thread::hardware_concurrency();
  size_t parts = 1;
  while (parts * 2 <= std::min<size_t>(hw ? hw : 1, 32) && n / (parts * 2) >= (1u << 16)) parts *= 2;
  if (parts == 1) {
    std::sort(idx.begin(), idx.end(), less);
    return;
  }
  std::vector<size_t> cut(parts + 1);
  for (size_t i = 0; i <= parts; ++i) cut[i] = n * i / parts;
  {
    std::vector<std::thread> th;
    for (size_t i = 0; i < parts; ++i)
      th.emplace_back([&, i] { std::sort(idx.begin() + cut[i], idx.begin() + cut[i + 1], less); });
    for (auto &t : th) t.join();
  }
  for (size_t width = 1; width < parts; width *= 2) {
    std::vector<std::thread> th;
    for (size_t i = 0; i + width < parts; i += 2 * width) {
      const size_t lo = cut[i], mid = cut[i + width], hi = cut[std::min(i + 2 * width, parts)];
      th.emplace_back([&, lo, mid, hi] { std::inplace_merge(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, less); });
    }
    for (auto &t : th) t.join();
  }
}

// fn(lo, hi) over [0, n) on several threads when n is large.
template <typename F>
void parallel_for(size_t n, F fn) {
  unsigned hw = std::thread::hardware_concurrency();
  const size_t parts = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 1, 32), n >> 16));
  if (parts == 1) {
    fn((size_t)0, n);
    return;
  }
  std::vector<std::thread> th;
  for (size_t i = 0; i < parts; ++i) th.emplace_back([&, i] { fn(n * i / parts, n * (i + 1) / parts); });
  for (auto &t : th) t.join();
}

// std::merge of two sorted index lists (no equal elements across them) on several threads:
// `a` is cut into equal runs, each run's first element finds its place in `b`, the pieces merge
// independently.  The compares chase ids all over the heap, so this is latency-bound work that
// scales with the cores.
template <typename Less>
void parallel_merge(const std::vector<uint32_t> &a, const std::vector<uint32_t> &b, std::vector<uint32_t> &out, Less less) {
  out.resize(a.size() + b.size());
  unsigned hw = std::thread::hardware_concurrency();
  const size_t parts = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 1, 32), a.size() >> 16));
  if (parts == 1) {
    std::merge(a.begin(), a.end(), b.begin(), b.end(), out.begin(), less);
    return;
  }
  std::vector<size_t> ca(parts + 1), cb(parts + 1);
  for (size_t i = 0; i <= parts; ++i) ca[i] = a.size() * i / parts;
  cb[0] = 0;
  cb[parts] = b.size();
  for (size_t i = 1; i < parts; ++i) cb[i] = (size_t)(std::lower_bound(b.begin(), b.end(), a[ca[i]], less) - b.begin());
  std::vector<std::thread> th;
  for (size_t i = 0; i < parts; ++i)
    th.emplace_back([&, i] {
      std::merge(a.begin() + ca[i], a.begin() + ca[i + 1], b.begin() + cb[i], b.begin() + cb[i + 1],
                 out.begin() + ca[i] + cb[i], less);
    });
  for (auto &t : th) t.join();
}

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t count = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    count = 0;
  }
  int ensure(size_t want) {
    if (want <= count) return VT_OK;
    release();
    VT_HIP(hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T)));
    count = want;
    return VT_OK;
  }
};

template <typename T>
struct PinnedBuf {
  T *p = nullptr;
  size_t count = 0;
  ~PinnedBuf() { release(); }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    count = 0;
  }
  int ensure(size_t want) {
    if (want <= count) return VT_OK;
    release();
    VT_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), want * sizeof(T), hipHostMallocMapped));
    count = want;
    return VT_OK;
  }
};

// The row matrix of a shard.
//  * small: one hipMalloc, regrown by allocate + copy (cheap below a chunk);
//  * from one chunk on: ONE reserved virtual range as large as the card's memory, physical
//    chunks of equal size (1 GiB) mapped behind each other as the rows arrive
//    (hipMemAddressReserve / hipMemCreate / hipMemMap).  Growing maps more chunks: no copy, no
//    second slab beside the first (regrowing a 100-GB slab by allocate + copy needs 300 GB for a
//    moment -- more than the card has), the rows never move.  Streaming over a mapped range
//    runs at the rate of a hipMalloc'ed one (tools/vmm_probe.hip: the +-2 % between two
//    allocations of either kind is placement luck).  Chunks of one range must be equally
//    large: hipMemSetAccess rejects most mixed sequences on ROCm 7.
struct Slab {
  float *p = nullptr;
  size_t bytes = 0;     // usable bytes behind p
  size_t defined = 0;   // bytes that hold rows or zeros (chunks mapped by a growth that failed later are not, yet)
  bool mapped = false;  // p is a reserved range with `chunks` mapped at its start
  size_t reserved = 0, chunk = 0;
  std::vector<hipMemGenericAllocationHandle_t> chunks;

  ~Slab() { release(); }
  void release() {
    if (mapped) {
      if (bytes) (void)hipMemUnmap(p, bytes);
      for (auto h : chunks) (void)hipMemRelease(h);
      if (p) (void)hipMemAddressFree(p, reserved);
    } else if (p) {
      (void)hipFree(p);
    }
    p = nullptr;
    bytes = defined = reserved = chunk = 0;
    mapped = false;
    chunks.clear();
  }
  static size_t chunk_bytes() {
    const char *e = std::getenv("VT_SLAB_CHUNK_MB");  // (tests: small chunks, so that small corpora cross chunk borders)
    const long mb = e ? std::atol(e) : 0;
    return mb > 0 ? (size_t)mb << 20 : (size_t)1 << 30;
  }
  static bool mapping_allowed() {
    const char *e = std::getenv("VT_SLAB");
    return !(e && std::strcmp(e, "malloc") == 0);
  }
  // Maps chunks until `want` bytes are usable.  Failure leaves what was mapped before intact.
  int map_up_to(size_t want, int device) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    while (bytes < want) {
      if (bytes + chunk > reserved) return fail(VT_ERR_DEVICE, "row slab: the reserved range is exhausted");
      hipMemGenericAllocationHandle_t h;
      VT_HIP(hipMemCreate(&h, chunk, &prop, 0));
      char *at = reinterpret_cast<char *>(p) + bytes;
      hipError_t e = hipMemMap(at, chunk, 0, h, 0);
      if (e == hipSuccess) {
        e = hipMemSetAccess(at, chunk, &acc, 1);
        if (e != hipSuccess) (void)hipMemUnmap(at, chunk);
      }
      if (e != hipSuccess) {
        (void)hipMemRelease(h);
        return fail(VT_ERR_DEVICE, std::string("row slab: ") + hipGetErrorString(e));
      }
      chunks.push_back(h);
      bytes += chunk;
    }
    return VT_OK;
  }
  // A fresh mapped slab of at least `want` bytes (nothing copied).  VT_ERR_UNSUPPORTED when the
  // runtime has no virtual memory management (the caller stays with hipMalloc).
  int start_mapped(size_t want, int device) {
    release();
    size_t free_b = 0, total_b = 0;
    VT_HIP(hipMemGetInfo(&free_b, &total_b));
    chunk = chunk_bytes();
    reserved = (std::max(total_b, want) + chunk - 1) / chunk * chunk;
    void *base = nullptr;
    if (hipMemAddressReserve(&base, reserved, 0, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError();
      reserved = chunk = 0;
      return VT_ERR_UNSUPPORTED;
    }
    p = reinterpret_cast<float *>(base);
    mapped = true;
    const int st = map_up_to(want, device);
    if (st != VT_OK) release();
    return st;
  }
};

using vt::ResultBlock;

// Per-device execution context: stream, scratch, profiling.
struct Ctx {
  int device = 0;
  int num_cus = 256;
  int blocks_per_cu = 2;
  int hamming_blocks_per_cu = 2;
  hipStream_t stream = nullptr;
  DevBuf<float> dQ;
  uint64_t *dQbits = nullptr;  // the query's sign bits, behind the query in dQ (upload_query with_bits)
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
  uint32_t begin_rows = 0, begin_dim = 0;  // scan of the last vt_flat_search_begin (profiling)
  bool profiling = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  vt_profile prof{};

  ~Ctx() {
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev2) (void)hipEventDestroy(ev2);
    if (ev3) (void)hipEventDestroy(ev3);
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
    if (const char *e = std::getenv("VT_HAMMING_BLOCKS_PER_CU")) {
      const int v = std::atoi(e);
      if (v >= 1 && v <= 8) hamming_blocks_per_cu = v;
    }
    if (const char *e = std::getenv("VT_BLOCKS_PER_CU")) {
      const int v = std::atoi(e);
      if (v >= 1 && v <= 8) blocks_per_cu = v;
    }
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

struct vt_hits {
  std::vector<std::string> ids;
  std::vector<float> raw;
  std::vector<uint32_t> rank_key;
};

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
  int order = g_default_order;
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
  double max_sqnorm = -1.0;  // max_i sum_j x_ij^2, < 0 = stale (error margin of the batched path)
  DevBuf<float> dXnorm2;     // per-row squared norms, valid with max_sqnorm
  // ids
  std::vector<std::string> ids;  // by row
  std::unordered_map<std::string, uint32_t> row_of;
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

  ~Shard() {
    (void)hipSetDevice(ctx.device);
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
// kMaxContexts), else wait.  Held only under the handle's shared lock.
struct CtxLease {
  Shard *ix;
  Ctx *c = nullptr;
  int status = VT_OK;
  explicit CtxLease(Shard *s) : ix(s) {
    std::unique_lock<std::mutex> g(ix->pool_mu);
    for (;;) {
      if (!ix->ctx0_busy) {
        ix->ctx0_busy = true;
        c = &ix->ctx;
        return;
      }
      if (!ix->free_ctx.empty()) {
        c = ix->free_ctx.back();
        ix->free_ctx.pop_back();
        return;
      }
      if (ix->extra.size() + 1 < kMaxContexts) {
        auto nc = std::make_unique<Ctx>();
        status = nc->init(ix->ctx.device);
        if (status != VT_OK) return;
        nc->profiling = ix->ctx.profiling;
        c = nc.get();
        ix->extra.push_back(std::move(nc));
        return;
      }
      ix->pool_cv.wait(g);
    }
  }
  ~CtxLease() {
    if (!c) return;
    {
      std::lock_guard<std::mutex> g(ix->pool_mu);
      if (c == &ix->ctx) ix->ctx0_busy = false;
      else ix->free_ctx.push_back(c);
    }
    ix->pool_cv.notify_one();
  }
  CtxLease(const CtxLease &) = delete;
  CtxLease &operator=(const CtxLease &) = delete;
};

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

// One thread per shard of a multi-shard index, bound to the shard's device: the caller
// posts the same job to all of them, so the launch overheads of the shards overlap and
// each shard's kernels are issued by a thread whose current device never changes.
struct Worker {
  struct Job {
    std::function<int()> fn;
    int status = VT_OK;
    std::string error;
    bool done = false;
  };
  std::thread th;
  std::mutex mu;
  std::condition_variable cv, done_cv;
  std::deque<Job *> queue;
  bool stop = false;
  int device = 0;

  void start(int dev) {
    device = dev;
    th = std::thread([this] { loop(); });
  }
  void loop() {
    (void)hipSetDevice(device);
    for (;;) {
      Job *job = nullptr;
      {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [this] { return stop || !queue.empty(); });
        if (queue.empty()) return;  // stop
        job = queue.front();
        queue.pop_front();
      }
      g_last_error.clear();
      const int st = guarded(job->fn);
      {
        std::lock_guard<std::mutex> g(mu);
        job->status = st;
        if (st != VT_OK) job->error = g_last_error;
        job->done = true;
      }
      done_cv.notify_all();
    }
  }
  void post(Job *job) {
    {
      std::lock_guard<std::mutex> g(mu);
      queue.push_back(job);
    }
    cv.notify_one();
  }
  void wait(Job *job) {
    std::unique_lock<std::mutex> g(mu);
    done_cv.wait(g, [job] { return job->done; });
  }
  ~Worker() {
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
    }
    cv.notify_one();
    if (th.joinable()) th.join();
  }
};

}  // namespace

// The handle behind the C ABI: FlatResource(RwLock<FlatIndex>) (flat.rs:13-17, nifs.rs:254-257).
struct vt_flat {
  // searches share, mutations exclude (nifs.rs:266-309)
  mutable std::shared_mutex rw;
  // A mutation that failed on the device after it had begun changing the index leaves it
  // poisoned, like a panic under the reference's write lock: every later call fails with
  // "flat lock poisoned" (nifs.rs:269).
  bool poisoned = false;
  int metric = 0;
  long dim = -1;  // FlatIndex.dimension across all shards
  std::vector<std::unique_ptr<Shard>> shards;
  // multi-shard only
  std::vector<std::unique_ptr<Worker>> workers;
  std::mutex post_mu;  // jobs reach every worker's queue in one order (collectives must match up)
  int exchange = VT_EXCHANGE_HOST;
  bool exchange_forced = false;
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
  struct Waiting {
    const float *query;
    size_t n, limit;
    vt_hits **out;
    int status = VT_OK;
    std::string error;
    enum { QUEUED, LEADS, ALONE, DONE } state = QUEUED;
    std::condition_variable wake;  // its own: a finished batch wakes exactly its members and the next leader
    Waiting(const float *q, size_t n_, size_t limit_, vt_hits **out_) : query(q), n(n_), limit(limit_), out(out_) {}
  };
  struct Coalescer {
    std::mutex mu;
    std::condition_variable gather;  // a leader waiting a moment for the callers it expects back
    std::deque<Waiting *> waiting;
    unsigned active = 0;        // searches / batches running
    size_t last_batch = 1;      // members of the last batch that ran
    double last_seconds = 0.0;  // what it took
    uint64_t batches = 0, batched_queries = 0;
  } co;
  std::atomic<uint64_t> approx_bytes{0};  // rows x row stride, refreshed by mutations (the coalescer's only use of it is a size class)

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

namespace {

// Internal (never crosses the ABI): a device-side list overflowed, redo on the general path.
constexpr int kRetryInternal = -100;

// Rough per-call times on MI355X (tools/size_probe.py, tools/latency_floor.py), used only
// to choose between equivalent code paths: one fused scan of `bytes`, and the fixed cost
// the multi-kernel paths add on top of their scans.
constexpr double kScanFixedS = 35e-6, kScanBytesPerS = 6.5e12;
constexpr double kThresholdFixedS = 140e-6, kBatchFixedS = 180e-6, kBatchFlopsPerS = 135e12;
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
  if (total <= (size_t)vt::kMaxFusedK || total > (size_t)vt::kSelListMax || n < kThresholdMinRows ||
      std::getenv("VT_NO_THRESHOLD_SELECT"))
    return false;
  if (std::getenv("VT_FORCE_THRESHOLD_SELECT")) return true;  // tests: exercise the path on small corpora
  const double passes = std::ceil((double)total / vt::kMaxFusedK);
  const double t_pass = pass_fixed_s + scan_bytes / kScanBytesPerS;
  const double t_loop = passes * t_pass;
  const double t_threshold = t_pass + kThresholdFixedS + passes * 25e-6;
  return t_threshold < t_loop;
}

// Limits above kSelListMax (flat.ex:98-103 allows up to 2^32 - 1) in ONE scan: key and payload
// columns, the device-side radix threshold, and the collected list -- every key up to the k-th
// one's 33-bit prefix, a few more than k -- handed to the host as it is, which cuts and orders it
// (nth_element + sort of ~k entries).  kRetryInternal: more ties at the threshold than the list
// holds; the caller takes the pass-per-256 loop.
int threshold_big(Ctx &c, const vt::ScanArgs &scan, uint32_t blocks, uint32_t n, uint32_t k, bool timed, uint32_t d,
                  std::vector<vt::Entry> &out) {
  VT_TRY(c.dKeyCol.ensure(((size_t)n + 1) / 2 * 2));
  VT_TRY(c.dPayCol.ensure(n));
  VT_TRY(c.dRadixHist.ensure(3 * vt::kRadixBins));
  VT_TRY(c.dRadixCount.ensure(1));
  VT_TRY(c.dPartKeys.ensure(kThresholdListCap));
  VT_TRY(c.dPartPay.ensure(kThresholdListCap));
  VT_TRY(c.hListKeys.ensure(kThresholdListCap));
  VT_TRY(c.hListPay.ensure(kThresholdListCap));
  vt::ScanArgs a = scan;
  a.k = 1;
  a.key_out = c.dKeyCol.p;
  a.pay_out = c.dPayCol.p;
  if (timed) VT_HIP(hipEventRecord(c.ev0, c.stream));
  VT_HIP(vt::launch_scan(a, blocks, c.stream));
  if (timed) VT_HIP(hipEventRecord(c.ev1, c.stream));
  VT_HIP(hipMemsetAsync(c.dRadixHist.p, 0, 3 * vt::kRadixBins * sizeof(uint32_t), c.stream));
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
  r.pay_col = c.dPayCol.p;
  const uint32_t rblocks = (uint32_t)c.num_cus * 8;
  for (int pass = 0; pass < 3; ++pass) VT_HIP(vt::launch_radix_pass(r, pass, rblocks, c.stream));
  VT_HIP(vt::launch_radix_collect(r, rblocks, c.stream));
  uint32_t count = 0;
  int status = 0;
  VT_HIP(hipMemcpyAsync(&count, c.dRadixCount.p, sizeof(count), hipMemcpyDeviceToHost, c.stream));
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
  if (status == vt::kStatusRetry || count > kThresholdListCap) return kRetryInternal;
  if (status == VT_ERR_OVERFLOW) return VT_ERR_OVERFLOW;
  VT_HIP(hipMemcpyAsync(c.hListKeys.p, c.dPartKeys.p, (size_t)count * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipMemcpyAsync(c.hListPay.p, c.dPartPay.p, (size_t)count * sizeof(vt::Payload), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  std::vector<vt::Entry> list(count);
  for (uint32_t i = 0; i < count; ++i) {
    list[i].key = c.hListKeys.p[i];
    list[i].row = c.hListPay.p[i].row;
    list[i].raw = c.hListPay.p[i].raw;
  }
  const size_t take = std::min<size_t>(k, list.size());
  auto by_key = [](const vt::Entry &x, const vt::Entry &y) { return x.key < y.key; };
  std::nth_element(list.begin(), list.begin() + (take ? take - 1 : 0), list.end(), by_key);
  std::sort(list.begin(), list.begin() + take, by_key);
  out.assign(list.begin(), list.begin() + take);
  return VT_OK;
}

int threshold_rows(Ctx &c, uint32_t n, uint32_t k) {
  VT_TRY(c.dRadixHist.ensure(3 * vt::kRadixBins));
  VT_TRY(c.dRadixCount.ensure(1));
  VT_TRY(c.dPartKeys.ensure(kThresholdListCap));
  VT_TRY(c.dPartPay.ensure(kThresholdListCap));
  VT_TRY(c.dListKeys.ensure(k));
  VT_TRY(c.dListPay.ensure(k));
  VT_HIP(hipMemsetAsync(c.dRadixHist.p, 0, 3 * vt::kRadixBins * sizeof(uint32_t), c.stream));
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
  const uint32_t blocks = (uint32_t)c.num_cus * 8;
  for (int pass = 0; pass < 3; ++pass) VT_HIP(vt::launch_radix_pass(r, pass, blocks, c.stream));
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
  if (j.metric == VT_JACCARD && j.d >= 4096)
    return fail(VT_ERR_UNSUPPORTED, "jaccard on device supports d < 4096");
  const uint32_t tile_rows = vt::scan_tile_rows(j.n, j.d, c.resident_waves());
  const uint32_t ntiles = (j.n + tile_rows - 1) / tile_rows;
  // very wide rows leave no LDS for the large candidate buffer: smaller passes
  const size_t kmax = vt::scan_lds_bytes(j.d, vt::kMaxFusedK) ? (size_t)vt::kMaxFusedK : (size_t)vt::kSmallK;
  uint64_t lo = 0;
  bool has_lo = false;
  const size_t total = std::min<size_t>(want, j.n);
  if (!j.gather && out.empty() && total > (size_t)vt::kSelListMax && total <= (size_t)kThresholdListCap &&
      j.n >= kThresholdMinRows && !std::getenv("VT_NO_THRESHOLD_SELECT")) {
    vt::ScanArgs a{};
    a.X = j.X;
    a.stride = j.stride;
    a.q = c.dQ.p;
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
    a.q = c.dQ.p;
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
    a.q = c.dQ.p;
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

int run_hamming(Ctx &c, const uint64_t *bits, const uint64_t *qbits, const uint32_t *id_rank, uint32_t n, uint32_t d,
                size_t want, std::vector<vt::Entry> &out, bool count_profile) {
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
int upload_query(Ctx &c, const float *q, size_t n, uint32_t *q_nonzero, bool with_bits = false) {
  const uint32_t ld = vt::padded_dim((uint32_t)n);
  const size_t words = (n + 63) / 64;
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
      if (q[i] >= 0.0f) w[i / 64] |= 1ull << (i % 64);
    c.dQbits = reinterpret_cast<uint64_t *>(c.dQ.p + ld);
  }
  VT_HIP(hipMemcpyAsync(c.dQ.p, c.hQ.p, total * sizeof(float), hipMemcpyHostToDevice, c.stream));
  return VT_OK;
}

inline uint32_t rank_key_of(uint64_t key) { return (uint32_t)(key >> 32); }

// ------------------------------------------------------------------ index ops
int index_reserve(Shard *ix, uint32_t want_rows) {
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
  ix->max_sqnorm = -1.0;
  ix->bits_dirty.clear();
  ix->norm_dirty.clear();
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
uint32_t index_row_for(Shard *ix, const char *id, size_t len, bool *is_new) {
  std::string key(id, len);
  auto it = ix->row_of.find(key);
  if (it != ix->row_of.end()) {
    *is_new = false;
    return it->second;
  }
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
  ix->row_of.emplace(key, r);
  ix->ids.push_back(std::move(key));
  return r;
}

// Recomputes id_rank (position of each row's id in bytewise order) if stale
// and makes the device copy current.
int index_sync_ranks(Shard *ix, bool force_upload) {
  if (!ix->ranks_clean) {
    // Rows that kept a rank from before are still in the right relative order
    // (ranks only need to be order-isomorphic to the ids): sort them by rank
    // (integers), sort only the unranked newcomers by id (strings), and merge.
    const std::vector<std::string> &ids = ix->ids;
    const std::vector<uint32_t> &rk = ix->rank_host;
    std::vector<uint32_t> ranked, fresh;
    uint32_t maxr = 0;
    size_t nranked = 0;
    for (uint32_t i = 0; i < ix->n; ++i)
      if (rk[i] != kUnranked) {
        maxr = std::max(maxr, rk[i]);
        ++nranked;
      }
    if (nranked && (uint64_t)maxr < 4ull * ix->n + 1024) {
      // ranks are unique: a bucket pass puts the ranked rows in rank (= id) order without sorting
      std::vector<uint32_t> slot((size_t)maxr + 1, kUnranked);
      for (uint32_t i = 0; i < ix->n; ++i)
        if (rk[i] != kUnranked) slot[rk[i]] = i;
      ranked.reserve(nranked);
      for (uint32_t v : slot)
        if (v != kUnranked) ranked.push_back(v);
      for (uint32_t i = 0; i < ix->n; ++i)
        if (rk[i] == kUnranked) fresh.push_back(i);
    } else {
      ranked.reserve(ix->n);
      for (uint32_t i = 0; i < ix->n; ++i) (rk[i] == kUnranked ? fresh : ranked).push_back(i);
      parallel_sort(ranked, [&rk](uint32_t a, uint32_t b) { return rk[a] < rk[b]; });
    }
    parallel_sort(fresh, [&ids](uint32_t a, uint32_t b) { return ids[a] < ids[b]; });
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
  }
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
};

constexpr size_t kMaxDerivedDirty = 65536;  // more mutated rows than this: rebuild instead of patching

// Row `r` changed: its sign bits and norm are stale.
inline void index_touch_row(Shard *ix, uint32_t r) {
  if (ix->bits_valid) {
    ix->bits_dirty.push_back(r);
    if (ix->bits_dirty.size() > kMaxDerivedDirty) {
      ix->bits_valid = false;
      ix->bits_dirty.clear();
    }
  }
  if (ix->max_sqnorm >= 0.0) {
    ix->norm_dirty.push_back(r);
    if (ix->norm_dirty.size() > kMaxDerivedDirty) {
      ix->max_sqnorm = -1.0;
      ix->norm_dirty.clear();
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

// Shared body of insert / insert_many / load_matrix: rows are already validated.
// `*began` is set once the index has started to change: a failure after that point
// leaves it inconsistent (the caller poisons the handle).
int index_store_rows(Shard *ix, size_t count, const char *ids, const size_t *id_off, const RowSource &src, bool *began) {
  if (count == 0) return VT_OK;
  Ctx &c = ix->ctx;
  const size_t d = (size_t)ix->dim;
  if ((uint64_t)ix->n + count > 0xFFFFFFF0ull) return fail(VT_ERR_UNSUPPORTED, "more than 2^32-16 rows");
  VT_TRY(index_reserve(ix, ix->n + (uint32_t)count));
  const uint32_t n_before = ix->n;
  std::vector<uint32_t> target(count);
  bool all_appended_in_order = true;
  if (count > 1024) {
    // bulk load: no rehash / regrowth inside the id loop -- but geometric, or a corpus that arrives
    // in many appends re-hashes and re-copies its whole id table at every one of them (84 M ids: 15 s)
    const size_t need = (size_t)ix->n + count;
    if ((double)need > (double)ix->row_of.bucket_count() * ix->row_of.max_load_factor())
      ix->row_of.reserve(std::max(need, 2 * ix->row_of.size()));
    if (need > ix->ids.capacity()) ix->ids.reserve(std::max(need, 2 * ix->ids.capacity()));
    if (need > ix->rank_host.capacity()) ix->rank_host.reserve(std::max(need, 2 * ix->rank_host.capacity()));
  }
  *began = true;
  for (size_t i = 0; i < count; ++i) {
    bool is_new = false;
    target[i] = index_row_for(ix, ids + id_off[i], id_off[i + 1] - id_off[i], &is_new);
    if (!is_new || target[i] != n_before + i) all_appended_in_order = false;
  }
  // (test hook: a device failure between the id table's change and the rows' arrival, the one
  // window in which a mutation cannot be taken back -- tests/test_gpu_multishard.py checks that
  // the handle is poisoned from then on)
  if (std::getenv("VT_TEST_FAIL_AFTER_ID_UPDATE")) return fail(VT_ERR_DEVICE, "injected failure after the id table changed");
  if (count > kMaxDerivedDirty) {
    ix->bits_valid = false;
    ix->max_sqnorm = -1.0;
    ix->bits_dirty.clear();
    ix->norm_dirty.clear();
  } else {
    for (size_t i = 0; i < count; ++i) index_touch_row(ix, target[i]);
  }
  const uint32_t ld = ix->ld;
  bool pending = false;
  if (src.device) {
    bool picks_dense = true;  // the batch is one contiguous block of the source
    if (src.pick)
      for (size_t i = 1; i < count && picks_dense; ++i) picks_dense = src.pick[i] == src.pick[0] + i;
    const float *first = src.device + (src.pick ? (size_t)src.pick[0] * d : 0);
    // (test hook: a one-GPU box has no other device to own the rows)
    const bool foreign = device_of_pointer(src.device) != c.device || std::getenv("VT_TEST_FOREIGN_ROWS") != nullptr;
    if (foreign && ix->slab.mapped) {
      // Rows that live on another device of the node, bound for a mapped slab: only this device
      // has been given access to the slab's chunks (hipMemSetAccess), so a peer copy must not
      // target it.  The rows cross into an ordinary buffer here first (blocks of <= 256 MB), and
      // are placed from there by local copies.
      const size_t block_rows = std::max<size_t>(1, ((size_t)256 << 20) / (d * sizeof(float)));
      DevBuf<float> stage;
      VT_TRY(stage.ensure(std::min(count, block_rows) * d));
      size_t i = 0;
      while (i < count) {
        // a run of consecutive source rows, at most one block long
        size_t e = i + 1;
        const size_t p0 = src.pick ? src.pick[i] : i;
        while (e < count && e - i < block_rows && (src.pick ? src.pick[e] : e) == p0 + (e - i)) ++e;
        VT_HIP(hipMemcpyAsync(stage.p, src.device + p0 * d, (e - i) * d * sizeof(float), hipMemcpyDefault, c.stream));
        for (size_t j = i; j < e;) {  // ... placed in runs of consecutive slab rows
          size_t r = j + 1;
          while (r < e && target[r] == target[j] + (uint32_t)(r - j)) ++r;
          float *dst = ix->dX + (size_t)target[j] * ld;
          if (ld == d) VT_HIP(hipMemcpyAsync(dst, stage.p + (j - i) * d, (r - j) * d * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
          else VT_HIP(vt::launch_pad_rows(stage.p + (j - i) * d, (uint32_t)(r - j), (uint32_t)d, dst, ld, c.stream));
          j = r;
        }
        VT_HIP(hipStreamSynchronize(c.stream));  // the block is reused
        i = e;
      }
    } else if (all_appended_in_order && picks_dense) {
      // (hipMemcpyDefault: the source may live on another device of the node)
      float *dst = ix->dX + (size_t)n_before * ld;
      if (ld == d) VT_HIP(hipMemcpyAsync(dst, first, count * d * sizeof(float), hipMemcpyDefault, c.stream));
      else VT_HIP(vt::launch_pad_rows(first, (uint32_t)count, (uint32_t)d, dst, ld, c.stream));
    } else if (count < 64 || foreign) {
      // (few rows, or rows that live on another device: plain copies, which need no peer mapping)
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
  const bool rank_now = !ix->ranks_clean && count >= kBulkRankRows && count >= (size_t)n_before / 4;
  if (!ix->ranks_clean && !rank_now) {
    // the device column is brought up to date lazily (index_lazy_ranks) or by the next re-rank
    if (count < kBulkRankRows)
      for (uint32_t r = n_before; r < ix->n; ++r) ix->rank_dirty.push_back(r);
    else
      ix->rank_dirty_all = true;
    if (ix->rank_dirty.size() > kMaxDirtyRanks) ix->rank_dirty_all = true;
  }
  // keep device ranks current when they stayed valid (sorted appends)
  if (ix->ranks_clean && ix->n > n_before) {
    uint32_t from = n_before;
    if (ix->dRank.count < ix->cap) {
      VT_TRY(ix->dRank.ensure(ix->cap));
      from = 0;
    }
    VT_HIP(hipMemcpyAsync(ix->dRank.p + from, ix->rank_host.data() + from, (size_t)(ix->n - from) * sizeof(uint32_t),
                          hipMemcpyHostToDevice, c.stream));
    pending = true;
  }
  if (pending) VT_HIP(hipStreamSynchronize(c.stream));
  if (rank_now) VT_TRY(index_sync_ranks(ix, false));
  return VT_OK;
}

int make_hits(const Shard *ix, const std::vector<vt::Entry> &entries, vt_hits **out) {
  auto h = std::make_unique<vt_hits>();
  h->ids.reserve(entries.size());
  for (const auto &e : entries) {
    h->ids.push_back(ix->ids[e.row]);
    h->raw.push_back(e.raw);
    h->rank_key.push_back(rank_key_of(e.key));
  }
  *out = h.release();
  return VT_OK;
}

int empty_hits(vt_hits **out) {
  *out = new vt_hits();
  return VT_OK;
}

// ---- what a reader needs up to date before it may run under the shared lock ----------
enum : unsigned { NEED_RANKS = 1, NEED_STRICT_RANKS = 2, NEED_BITS = 4, NEED_NORMS = 8 };
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
         ix->unranked <= std::max<size_t>(65536, ix->n / 8) && !std::getenv("VT_EAGER_RANKS");
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
  if ((need & NEED_NORMS) && (ix->max_sqnorm < 0.0 || !ix->norm_dirty.empty() || ix->dXnorm2.count < rows)) return true;
  return false;
}

int index_ensure_bits(Shard *ix);
int index_ensure_norms(Shard *ix);

// Brings the derived columns a reader needs up to date (exclusive access; primary context).
int shard_prepare(Shard *ix, unsigned need, size_t limit) {
  if (ix->n == 0) return VT_OK;
  if (need & (NEED_RANKS | NEED_STRICT_RANKS)) {
    if (!(need & NEED_STRICT_RANKS) && lazy_ranks_ok(ix, limit)) VT_TRY(index_lazy_ranks(ix));
    else VT_TRY(index_sync_ranks(ix, false));
  }
  if (need & NEED_BITS) VT_TRY(index_ensure_bits(ix));
  if (need & NEED_NORMS) VT_TRY(index_ensure_norms(ix));
  return VT_OK;
}

// flat.rs:96-124 on a shard whose rank column shard_prepare has brought up to date --
// strictly (ranks_clean) or lazily (newcomers share kUnranked).  Read-only on the shard.
int search_ready(Shard *ix, Ctx &c, const float *query, size_t n, size_t limit, vt_hits **out) {
  if (limit == 0) return empty_hits(out);
  VT_TRY(validate_vector(query, n, ix->dim));
  if (ix->n == 0) return empty_hits(out);
  const bool lazy = !ix->ranks_clean;
  const size_t lazy_want = std::min<size_t>(limit, ix->n) + (limit < ix->n ? 1 : 0);
  uint32_t qnz = 0;
  VT_TRY(upload_query(c, query, n, &qnz));
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
  if (lazy) {
    // one hit more than asked for: if it does not tie with the last wanted one, the set is
    // exact whatever the unranked rows' id order is, and equal-rank runs are put in id order here
    VT_TRY(run_scan(c, j, lazy_want, entries, true));
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
  VT_TRY(run_scan(c, j, limit, entries, true));
  return make_hits(ix, entries, out);
}

// The same for a caller that owns the shard outright (a shard worker, or any caller under
// the exclusive lock): prepare, run on the primary context, settle a boundary tie.
int search_owner(Shard *ix, const float *query, size_t n, size_t limit, vt_hits **out) {
  if (shard_stale(ix, NEED_RANKS, limit)) VT_TRY(shard_prepare(ix, NEED_RANKS, limit));
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
    a.q = c.dQ.p;
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
    g.q = c.dQ.p;
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
    a.q = c.dQ.p;
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
    a.q = c.dQ.p;
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

// One funnel / rerank stage that never leaves the device: scores `count` rows
// (all rows, or the Entry.row column of the previous stage's block), keeps
// `want` <= kMaxFusedK of them in `dst`.  Nothing is waited for; an overflow
// flag raised by any stage stays in c.dStatus until a select with `last` moves
// it into its block.
int funnel_stage_dev(Shard *ix, Ctx &c, const float *query, uint32_t d, const ResultBlock *src, uint32_t count,
                     uint32_t want, uint32_t qnz, ResultBlock *dst, bool last) {
  const uint32_t *gather = src ? &src->e[0].row : nullptr;
  const uint32_t gstride = sizeof(vt::Entry) / sizeof(uint32_t);
  int *status = last ? c.dStatus.p : nullptr;
  if (ix->metric == VT_COSINE) {
    double qq = 0.0;  // f64_dot(q, q) over the prefix (distances.rs:179-185)
    for (uint32_t j = 0; j < d; ++j) qq += (double)query[j] * (double)query[j];
    if (!src) {
      const uint32_t blocks = c.grid_for((ix->n + 63) / 64, vt::cosine_scan_lds_bytes(d, want));
      VT_TRY(c.dPartKeys.ensure((size_t)blocks * want));
      VT_TRY(c.dPartPay.ensure((size_t)blocks * want));
      vt::CosineScanArgs a{};
      a.X = ix->dX;
      a.stride = ix->ld;
      a.q = c.dQ.p;
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
    a.q = c.dQ.p;
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
  const uint32_t tile_rows = vt::scan_tile_rows(count, d, c.resident_waves());
  const uint32_t ntiles = (count + tile_rows - 1) / tile_rows;
  const uint32_t blocks = c.grid_for(ntiles, vt::scan_lds_bytes(d, want));
  const uint32_t lists = vt::scan_lists(blocks);
  VT_TRY(c.dPartKeys.ensure((size_t)lists * want));
  VT_TRY(c.dPartPay.ensure((size_t)lists * want));
  vt::ScanArgs a{};
  a.X = ix->dX;
  a.stride = ix->ld;
  a.q = c.dQ.p;
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
  VT_HIP(vt::launch_scan(a, blocks, c.stream));
  VT_HIP(vt::launch_select(c.dPartKeys.p, c.dPartPay.p, lists * want, want, 0, 0, status, dst, c.dSelKeys.p, c.dSelPay.p,
                           c.stream));
  return VT_OK;
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

// Sign bits of every stored row in K4's layout, built on first use.
int index_ensure_bits(Shard *ix) {
  Ctx &c = ix->ctx;
  const uint32_t d = (uint32_t)ix->dim, words = (d + 63) / 64;
  // compress_sign_bits of every stored row (collection.ex:926): kept in HBM
  const size_t bwords = vt::hamming_matrix_words(std::max<uint32_t>(ix->cap, ix->n), words);
  if (ix->bits_valid && ix->dBits.count >= bwords) {
    // only the rows mutated since the last use
    uint32_t count = 0;
    VT_TRY(upload_row_list(ix, ix->bits_dirty, &count));
    VT_HIP(vt::launch_sign_pack_rows(ix->dX, ix->ld, c.dRankPairs.p, count, d, ix->dBits.p, c.stream));
    if (count) VT_HIP(hipStreamSynchronize(c.stream));  // the pinned list is reused by the next caller
    ix->bits_dirty.clear();
    return VT_OK;
  }
  ix->bits_dirty.clear();
  VT_TRY(ix->dBits.ensure(bwords));
  VT_HIP(hipMemsetAsync(ix->dBits.p, 0, bwords * sizeof(uint64_t), c.stream));
  VT_HIP(vt::launch_sign_pack(ix->dX, ix->ld, ix->n, d, ix->dBits.p, 1, c.stream));
  ix->bits_valid = true;
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

// ---------------------------------------------------------------- K2 host side
// One group of <= 256 queries through the matrix cores.  `done[i]` is set for
// every query whose exact top-k was proven complete; the others are left for
// the single-query path.
int batch_group(Shard *ix, Ctx &c, const float *queries, size_t nq, size_t limit, vt_hits **out, std::vector<char> &done) {
  const uint32_t d = (uint32_t)ix->dim, ld = ix->ld, n = ix->n;
  const uint32_t k = (uint32_t)std::min<size_t>(limit, n);
  uint32_t nq_pad = 32;
  while (nq_pad < nq) nq_pad *= 2;
  const uint32_t rows_per_block = vt::batch_rows_per_block(nq_pad);
  const uint32_t ntiles_total = (n + rows_per_block - 1) / rows_per_block;
  // pass-0 sample: 1/64 of the row tiles, 128..512 of them, spread over the corpus.
  // A larger sample gives a tighter tau: fewer candidates to rescore and, above
  // all, fewer trips through the epilogue's append path (a returning global
  // atomic, ~2 us with the matrix pipe idle: 5 % of the pass at 128 tiles).
  // (at most 65 536 sample rows: sample_tau_kernel holds a query's sample in registers)
  const uint32_t want_tiles = std::min<uint32_t>(std::min<uint32_t>(512, 65536 / rows_per_block),
                                                 std::max<uint32_t>(128, ntiles_total / 64));
  const uint32_t stride = std::max<uint32_t>(1, (ntiles_total + want_tiles - 1) / want_tiles);
  const uint32_t ntiles_sample = (ntiles_total + stride - 1) / stride;
  const uint32_t sample_rows = ntiles_sample * rows_per_block;
  // tau = rank-th best sample score: about rank * n / sample_rows rows pass
  const double ratio = (double)sample_rows / (double)n;
  uint32_t rank = (uint32_t)std::ceil(8.0 * k * std::min(1.0, ratio));
  rank = std::max<uint32_t>(3, std::min<uint32_t>(rank, std::min<uint32_t>(sample_rows, n)));
  const uint32_t cand_cap = 8192;
  constexpr uint32_t kBlocksPerQuery = 4;

  VT_TRY(c.dBQ.ensure((size_t)nq_pad * ld));
  VT_TRY(c.hBQ.ensure((size_t)nq_pad * ld));
  VT_TRY(c.dBTau.ensure(nq_pad));
  VT_TRY(c.hBTau.ensure(nq_pad));
  VT_TRY(c.dBSample.ensure((size_t)nq_pad * sample_rows));
  VT_TRY(c.dBCand.ensure((size_t)nq_pad * cand_cap));
  VT_TRY(c.dBCount.ensure(nq_pad));
  VT_TRY(c.hBCount.ensure(nq_pad));
  VT_TRY(c.dBOut.ensure((size_t)nq_pad * k));
  VT_TRY(c.hBOut.ensure((size_t)nq_pad * k));
  VT_TRY(c.dBOutCount.ensure(nq_pad));
  VT_TRY(c.hBOutCount.ensure(nq_pad));
  VT_TRY(c.dPartKeys.ensure((size_t)nq_pad * kBlocksPerQuery * k));
  VT_TRY(c.dPartPay.ensure((size_t)nq_pad * kBlocksPerQuery * k));

  std::vector<double> qnorm(nq);
  std::memset(c.hBQ.p, 0, (size_t)nq_pad * ld * sizeof(float));
  for (size_t i = 0; i < nq; ++i) {
    std::memcpy(c.hBQ.p + i * ld, queries + i * d, (size_t)d * sizeof(float));
    double s = 0.0;
    for (uint32_t j = 0; j < d; ++j) s += (double)queries[i * d + j] * (double)queries[i * d + j];
    qnorm[i] = std::sqrt(s);
  }
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
  // pass 0: dense scores of the sample -> tau
  a.n = sample_rows;
  a.sample_stride = stride;
  a.sample = c.dBSample.p;
  a.sample_rows = sample_rows;
  const uint32_t grid_cap = (uint32_t)c.num_cus;
  VT_HIP(vt::launch_batch_scores(a, true, std::min<uint32_t>(ntiles_sample, grid_cap), c.stream));
  VT_HIP(vt::launch_sample_tau(c.dBSample.p, sample_rows, nq_pad, (uint32_t)nq, rank, c.dBTau.p, c.stream));
  // pass 1: all rows, candidates with score >= tau
  a.n = n;
  a.sample = nullptr;
  a.tau = c.dBTau.p;
  a.cand = c.dBCand.p;
  a.cand_count = c.dBCount.p;
  a.cand_cap = cand_cap;
  VT_HIP(hipMemsetAsync(c.dBCount.p, 0, (size_t)nq_pad * sizeof(uint32_t), c.stream));
  if (c.profiling) VT_HIP(hipEventRecord(c.ev2, c.stream));
  VT_HIP(vt::launch_batch_scores(a, false, std::min<uint32_t>(ntiles_total, grid_cap), c.stream));
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
  VT_HIP(vt::launch_batch_select(c.dPartKeys.p, c.dPartPay.p, nq_pad, kBlocksPerQuery * k, k, c.dBOut.p, c.dBOutCount.p,
                                 c.stream));
  int status = 0;
  VT_HIP(hipMemcpyAsync(c.hBOut.p, c.dBOut.p, (size_t)nq_pad * k * sizeof(vt::Entry), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipMemcpyAsync(c.hBOutCount.p, c.dBOutCount.p, (size_t)nq_pad * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipMemcpyAsync(c.hBCount.p, c.dBCount.p, (size_t)nq_pad * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipMemcpyAsync(c.hBTau.p, c.dBTau.p, (size_t)nq_pad * sizeof(float), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipMemcpyAsync(&status, c.dStatus.p, sizeof(int), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipMemsetAsync(c.dStatus.p, 0, sizeof(int), c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  if (c.profiling) {
    float ms = 0.f;
    VT_HIP(hipEventElapsedTime(&ms, c.ev2, c.ev3));
    c.prof.batch_launches += 1;
    c.prof.batch_ms += ms;
    c.prof.batch_flops += 2.0 * (double)n * (double)nq_pad * (double)ld;
    c.prof.batch_queries += nq;
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
  const double u = std::ldexp(1.0, -24);
  const double xnorm = std::sqrt(ix->max_sqnorm);
  for (size_t i = 0; i < nq; ++i) {
    const uint32_t cnt = c.hBCount.p[i];
    if (cnt > cand_cap || c.hBOutCount.p[i] < k) continue;
    const vt::Entry *e = c.hBOut.p + i * k;
    const double tau = (double)c.hBTau.p[i];
    const double raw_k = (double)e[k - 1].raw;
    bool accept = false;
    if (l2_family) {
      const double eps = 3.5 * (double)d * u * (qnorm[i] + xnorm) * (qnorm[i] + xnorm);
      const double l2sq_k = (ix->metric == VT_L2 ? raw_k * raw_k : raw_k) * (1.0 + 16.0 * u);
      accept = l2sq_k <= qnorm[i] * qnorm[i] * (1.0 - 4.0 * u) - tau - eps;
    } else {
      const double eps = 2.5 * (double)d * u * qnorm[i] * xnorm;
      const double dot_k = ix->metric == VT_NEG_INNER_PRODUCT ? -raw_k : raw_k;
      const double slack = ix->metric == VT_COSINE ? 4.0 * u * std::max(1.0, std::fabs(1.0 - dot_k)) : 0.0;
      accept = tau + eps + slack <= dot_k;
    }
    if (!accept) continue;  // also taken when anything above is NaN
    std::vector<vt::Entry> entries(e, e + k);
    VT_TRY(make_hits(ix, entries, &out[i]));
    done[i] = 1;
  }
  return VT_OK;
}

// K1m serves a batch when every query's list fits its small wave buffers.
bool multi_scan_applies(const Shard *ix, size_t limit) {
  return limit >= 1 && std::min<size_t>(limit, ix->n) <= vt::scan_multi_max_k(vt::kMultiMaxQueries) &&
         !(ix->metric == VT_JACCARD && ix->dim >= 4096) && std::getenv("VT_NO_MULTI_SCAN") == nullptr;
}
double multi_scan_seconds(const Shard *ix, size_t nq) {
  const double sweeps = std::ceil((double)nq / vt::kMultiMaxQueries);
  const double waves = (double)ix->ctx.num_cus * 2 * vt::kWavesPerBlock;  // two blocks per CU
  const double tiles = std::ceil((double)ix->n / vt::scan_multi_tile_rows(vt::kMultiMaxQueries));
  double steps = std::ceil(tiles / waves) * std::ceil((double)ix->ld / 256.0);  // (tile, panel) steps of one wave
  double per_step = kMultiPanelS + 0.3e-6 / std::ceil((double)ix->ld / 256.0);
  if (ix->dim % 64 != 0) per_step *= 2.3;  // run-time bounds and lane order, compiler-scheduled loads
  return kMultiFixedS + sweeps * (kMultiSweepFixedS + steps * per_step + std::min(steps, 20.0) * kMultiRampS);
}

// `count` queries (rows `which[i]` of `queries`) in ceil(count / 8) sweeps of the corpus (K1m),
// every sweep and one batched select queued before the single wait.  Ranks strictly current.
int multi_scan_group(Shard *ix, Ctx &c, const float *queries, const std::vector<size_t> &which, size_t limit, vt_hits **out) {
  const uint32_t d = (uint32_t)ix->dim, ld = ix->ld, n = ix->n;
  const uint32_t k = (uint32_t)std::min<size_t>(limit, n);
  const size_t nq = which.size();
  const size_t lds = vt::scan_multi_lds_bytes(vt::kMultiMaxQueries);
  const uint32_t ntiles = (n + vt::scan_multi_tile_rows(vt::kMultiMaxQueries) - 1) / vt::scan_multi_tile_rows(vt::kMultiMaxQueries);
  const uint32_t blocks = c.grid_for(ntiles, lds);
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

// True when a batch of nq queries takes the shared MFMA pass (and so needs the row norms).
bool batch_uses_mfma(const Shard *ix, size_t nq, size_t limit) {
  const bool mfma_metric = ix->metric == VT_COSINE || ix->metric == VT_INNER_PRODUCT ||
                           ix->metric == VT_NEG_INNER_PRODUCT || ix->metric == VT_L2 || ix->metric == VT_L2_SQUARED;
  // one shared pass over the corpus costs about 1.3 single scans (HBM-bound below 33 queries),
  // so it pays from two queries on
  bool use_mfma = mfma_metric && nq >= 2 && limit <= (size_t)vt::kMaxFusedK && limit > 0 && ix->n >= 4096 &&
                  std::getenv("VT_BATCH_NO_MFMA") == nullptr;
  if (use_mfma && !std::getenv("VT_FORCE_BATCH_MFMA")) {  // (tests force the shared pass on small corpora)
    // nq single scans against one shared pass (HBM-bound below ~33 queries, then MFMA-bound)
    const double bytes = (double)ix->n * ix->ld * 4.0;
    double nq_pad = 32;
    while (nq_pad < (double)std::min<size_t>(nq, 256)) nq_pad *= 2;
    const double groups = std::ceil((double)nq / 256.0);
    const double t_pass = std::max(1.3 * bytes / kScanBytesPerS, 2.0 * ix->n * nq_pad * ix->ld / kBatchFlopsPerS);
    double t_other = (double)nq * scan_seconds(bytes);
    if (multi_scan_applies(ix, limit)) t_other = std::min(t_other, multi_scan_seconds(ix, nq));
    use_mfma = t_other > groups * (kBatchFixedS + t_pass);
  }
  return use_mfma;
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
  if (use_mfma) {
    for (size_t g0 = 0; g0 < nq; g0 += 256) {
      const size_t gn = std::min<size_t>(256, nq - g0);
      if (gn < 2) continue;  // a lone trailing query takes the single-query path below
      std::vector<char> gdone(gn, 0);
      VT_TRY(batch_group(ix, c, queries + g0 * d, gn, limit, out + g0, gdone));
      for (size_t i = 0; i < gn; ++i) done[g0 + i] = gdone[i];
    }
  }
  std::vector<size_t> left;
  for (size_t i = 0; i < nq; ++i)
    if (!done[i]) left.push_back(i);
  c.prof.batch_fallbacks += use_mfma ? left.size() : 0;
  // what the matrix cores did not take (no GEMM form for this metric, a small batch, a query
  // the bound could not certify): several queries per sweep of the corpus when their lists
  // fit, else one scan each
  if (left.size() >= 2 && multi_scan_applies(ix, limit) &&
      (multi_scan_seconds(ix, left.size()) < (double)left.size() * scan_seconds((double)ix->n * ix->ld * 4.0) ||
       std::getenv("VT_FORCE_MULTI_SCAN"))) {  // (tests force the sweep on corpora of a few thousand rows)
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
int quantized_ready(Shard *ix, Ctx &c, const float *query, size_t n, size_t candidates, size_t limit, vt_hits **out,
                    LocalStages *local = nullptr) {
  // collection.ex:276-295: prepare_query validates the query against the
  // collection's dimension; an empty store yields no candidates.
  VT_TRY(validate_vector(query, n, ix->dim));
  if (ix->n == 0 || candidates == 0 || limit == 0) return empty_hits(out);
  const uint32_t d = (uint32_t)ix->dim;
  const uint32_t words = (d + 63) / 64;
  uint32_t qnz = 0;
  VT_TRY(upload_query(c, query, n, &qnz, true));
  const size_t ncand = std::min<size_t>(candidates, ix->n);
  const size_t keep = local ? ncand : limit;
  // K4h needs integer bins in LDS, one fused select, and enough rows to be worth two passes
  const bool hist_ok = ncand <= (size_t)vt::kSelListMax && d <= vt::kHammingHistMaxDim &&
                       (ix->n >= 16384 || ncand > (size_t)vt::kMaxFusedK) &&
                       !std::getenv("VT_HAMMING_LISTS");
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
  if (use_hist) {
    // stage 1 as a pure stream (K4h): distance column + histogram, threshold collect,
    // select into the device block whose Entry.row column is stage 2's gather list
    const uint32_t k1 = (uint32_t)ncand;
    constexpr uint32_t kListCap = 65536, kHistStride = 8192;
    VT_TRY(c.dDist16.ensure(((size_t)std::max<uint32_t>(ix->cap, ix->n) + 7) / 8 * 8));
    VT_TRY(c.dHamHist.ensure(2 * kHistStride));
    VT_TRY(c.dHamCount.ensure(1));
    VT_TRY(c.dPartKeys.ensure(kListCap));
    VT_TRY(c.dPartPay.ensure(kListCap));
    VT_TRY(c.dStage.ensure(1));
    if (!c.ham_ready || c.ham_dirty) {
      VT_HIP(hipMemsetAsync(c.dHamHist.p, 0, 2 * kHistStride * sizeof(uint32_t), c.stream));
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
    h.hist = c.dHamHist.p + c.ham_parity * kHistStride;
    h.list_count = c.dHamCount.p;
    const uint32_t blocks = c.grid_for((ix->n + 63) / 64, vt::hamming_hist_lds_bytes(d), c.hamming_blocks_per_cu);
    if (c.profiling) VT_HIP(hipEventRecord(c.ev0, c.stream));
    VT_HIP(vt::launch_hamming_dist(h, blocks, c.stream));
    if (c.profiling) VT_HIP(hipEventRecord(c.ev1, c.stream));
    timed_hamming = c.profiling;
    vt::HammingCollectArgs g{};
    g.dist = c.dDist16.p;
    g.id_rank = ix->dRank.p;
    g.n = ix->n;
    g.d = d;
    g.k = k1;
    g.hist = h.hist;
    g.hist_next = c.dHamHist.p + (c.ham_parity ^ 1u) * kHistStride;
    g.list_count = c.dHamCount.p;
    g.keys = c.dPartKeys.p;
    g.pay = c.dPartPay.p;
    g.cap = kListCap;
    g.status = c.dStatus.p;
    VT_HIP(vt::launch_hamming_collect(g, (uint32_t)c.num_cus * 4, c.stream));
    c.ham_parity ^= 1u;
    c.ham_dirty = false;
    if (k1 <= (uint32_t)vt::kMaxFusedK) {
      // (no status pointer: a raised flag stays in dStatus for the final select)
      VT_HIP(vt::launch_select(c.dPartKeys.p, c.dPartPay.p, kListCap, k1, 0, 0, nullptr, c.dStage.p, c.dSelKeys.p, c.dSelPay.p,
                               c.stream, c.dHamCount.p));
      VT_TRY(copy_first_block());
      gather = &c.dStage.p->e[0].row;
      gather_stride = sizeof(vt::Entry) / sizeof(uint32_t);
    } else {
      if (local) return VT_ERR_ARGUMENT;  // callers keep one-round searches to candidates <= kMaxFusedK
      // up to 4 096 candidates (limit * 10 for limit <= 409): the exact candidate SET as a
      // device list -- stage 2 orders by its own keys, so this one need not be sorted
      VT_TRY(c.dListKeys.ensure(k1));
      VT_TRY(c.dListPay.ensure(k1));
      VT_HIP(vt::launch_select_list(c.dPartKeys.p, c.dPartPay.p, kListCap, c.dHamCount.p, k1, c.dListKeys.p, c.dListPay.p,
                                    c.stream));
      gather = &c.dListPay.p->row;
      gather_stride = sizeof(vt::Payload) / sizeof(uint32_t);
    }
  } else if (ncand <= (size_t)vt::kMaxFusedK) {
    // stage 1 stays on the device: hamming scan -> select into a device block
    // whose Entry.row column is the gather list of stage 2 (no host round trip)
    const uint32_t k1 = (uint32_t)ncand;
    const uint32_t blocks = c.grid_for((ix->n + 63) / 64, vt::hamming_lds_bytes(k1), c.hamming_blocks_per_cu);
    const uint32_t waves = vt::scan_lists(blocks);
    VT_TRY(c.dPartKeys.ensure((size_t)waves * k1));
    VT_TRY(c.dPartPay.ensure((size_t)waves * k1));
    VT_TRY(c.dStage.ensure(1));
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
    timed_hamming = c.profiling;
    VT_HIP(vt::launch_select(c.dPartKeys.p, c.dPartPay.p, waves * k1, k1, 0, 0, c.dStatus.p, c.dStage.p, c.dSelKeys.p, c.dSelPay.p, c.stream));
    VT_TRY(copy_first_block());
    gather = &c.dStage.p->e[0].row;
    gather_stride = sizeof(vt::Entry) / sizeof(uint32_t);
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
    a.q = c.dQ.p;
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
  VT_TRY(upload_query(c, query, n, &qnz_full));
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
  std::vector<Worker::Job> jobs(which.size());
  for (size_t i = 0; i < which.size(); ++i) {
    const size_t s = which[i];
    jobs[i].fn = [&fn, s]() -> int { return fn(s); };
  }
  {
    std::lock_guard<std::mutex> g(h->post_mu);
    for (size_t i = 0; i < which.size(); ++i) h->workers[which[i]]->post(&jobs[i]);
  }
  for (size_t i = 0; i < which.size(); ++i) h->workers[which[i]]->wait(&jobs[i]);
  for (size_t i = 0; i < which.size(); ++i)
    if (jobs[i].status != VT_OK) return fail(jobs[i].status, jobs[i].error);
  return VT_OK;
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
      VT_TRY(lease.c->bind());
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
    VT_TRY(shard_prepare(ix, need | NEED_STRICT_RANKS, limit));
    st = fn(ix, ix->ctx);
  }
  return st;
}

// flat.rs:88-93 on one shard.
int shard_delete(Shard *ix, const char *id, size_t id_len, bool *began) {
  Ctx &c = ix->ctx;
  auto it = ix->row_of.find(std::string(id ? id : "", id_len));
  if (it != ix->row_of.end()) {
    *began = true;
    ix->epoch += 1;
    const uint32_t r = it->second, last = ix->n - 1;
    if (ix->rank_host[r] == kUnranked && ix->unranked) ix->unranked -= 1;
    ix->row_of.erase(it);
    if (r != last) {
      // swap-delete: the last row moves into the hole and keeps its rank
      VT_HIP(hipMemcpyAsync(ix->dX + (size_t)r * ix->ld, ix->dX + (size_t)last * ix->ld, (size_t)ix->ld * sizeof(float),
                            hipMemcpyDeviceToDevice, c.stream));
      ix->ids[r] = std::move(ix->ids[last]);
      ix->row_of[ix->ids[r]] = r;
      ix->rank_host[r] = ix->rank_host[last];
      if (ix->ranks_clean && ix->dRank.p)
        VT_HIP(hipMemcpyAsync(ix->dRank.p + r, ix->dRank.p + last, sizeof(uint32_t), hipMemcpyDeviceToDevice, c.stream));
      if (!ix->ranks_clean) {
        ix->rank_dirty.push_back(r);
        if (ix->rank_dirty.size() > kMaxDirtyRanks) ix->rank_dirty_all = true;
      }
    }
    VT_HIP(hipMemsetAsync(ix->dX + (size_t)last * ix->ld, 0, (size_t)ix->ld * sizeof(float), c.stream));
    VT_HIP(hipStreamSynchronize(c.stream));
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
  a.q = c.dQ.p;
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
  for (const vt_hits *l : lists)
    if (l)
      for (size_t i = 0; i < l->ids.size(); ++i) items.push_back(MergeItem{l->rank_key[i], l->raw[i], &l->ids[i]});
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
      const ncclResult_t rc = r.AllGather(h->dBlock[s], h->dGather[s], bytes, ncclChar, h->comms[s], c.stream);
      if (rc != ncclSuccess) return fail(VT_ERR_DEVICE, std::string("ncclAllGather: ") + r.GetErrorString(rc));
      if (s == 0)
        VT_HIP(hipMemcpyAsync(h->hGather.p, h->dGather[0], S * bytes, hipMemcpyDeviceToHost, c.stream));
      VT_HIP(hipStreamSynchronize(c.stream));
      VT_TRY(settle_begin_profile(c));
      return st;
    }));
    std::vector<MergeItem> items;
    for (size_t s = 0; s < S; ++s) {
      const unsigned char *blk = h->hGather.p + s * bytes;
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
  int rc = on_all_shards(h, [&](size_t s) -> int {
    Shard *ix = h->shards[s].get();
    const unsigned need = NEED_STRICT_RANKS | (batch_uses_mfma(ix, nq, limit) ? NEED_NORMS : 0u);
    if (shard_stale(ix, need, limit)) VT_TRY(shard_prepare(ix, need, limit));
    return batch_ready(ix, ix->ctx, queries, nq, d, limit, per[s].data());
  });
  std::vector<vt_hits *> lists(S);
  for (size_t i = 0; i < nq && rc == VT_OK; ++i) {
    for (size_t s = 0; s < S; ++s) lists[s] = per[s][i];
    rc = merge_hit_lists(lists, limit, &out[i]);
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
    VT_TRY(upload_query(c, query, n, &qnz_full, kind == STAGE_HAMMING));
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
bool staged_one_round() { return std::getenv("VT_STAGED_ROUNDS") == nullptr; }  // (tests force the round-per-stage path)

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
    const int st = index_store_rows(ix, count, ids, id_off, src, &began);
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
    const int r = index_store_rows(ix, mine.size(), blob.data(), off.data(), sub, &b);
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
  uint64_t b = 0;
  for (auto &sh : h->shards) b += (uint64_t)sh->n * sh->ld * sizeof(float);
  h->approx_bytes.store(b, std::memory_order_relaxed);
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
  return read_single(h, NEED_RANKS, limit,
                     [&](Shard *ix, Ctx &c) -> int { return search_ready(ix, c, query, n, limit, out); });
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
    unsigned need = NEED_STRICT_RANKS;
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

// ---- searches that meet on one handle go together -----------------------------------------
// The reference's readers share an RwLock and scale with the host's cores.  Here every search
// is a pass over the corpus in HBM, and callers that run side by side on their own streams
// each read all of it.  So a search that finds another one running waits for it; whoever waits
// first then leads everything that has queued up with its limit as ONE batch -- the batch path
// gives every query the hits its own search would get, bit for bit -- and the others wake up
// with their lists.  An idle handle adds nothing: the first caller runs at once, alone.  Small
// corpora (latency-bound, not bandwidth-bound) keep two operations in flight.  What would force
// work a lone search avoids (a strict re-rank after unsorted inserts) is not batched: those
// callers are released to search side by side as before.  `VT_COALESCE=0` switches it off.
constexpr size_t kCoalesceMax = 256;
// Operations in flight before callers start to queue: a pass over a large corpus owns the
// memory system, two over a medium one still overlap their fixed costs, and searches of a
// corpus of a few MB are all fixed cost -- there every reader context runs side by side and
// only the callers beyond them travel together (tools/reader_probe.cpp).
unsigned coalesce_slots(uint64_t corpus_bytes) {
  if (const char *e = std::getenv("VT_COALESCE_SLOTS")) return (unsigned)std::max(1, std::atoi(e));
  return corpus_bytes < (64ull << 20) ? (unsigned)kMaxContexts : corpus_bytes < (1ull << 30) ? 2u : 1u;
}

bool coalescing_enabled() {
  const char *e = std::getenv("VT_COALESCE");
  return !(e && e[0] == '0');
}

// Runs the members of one batch (all with the leader's limit and query length).
void run_coalesced(vt_flat *h, std::vector<vt_flat::Waiting *> &members) {
  const size_t limit = members[0]->limit, n = members[0]->n;
  auto alone = [&](vt_flat::Waiting *w) {
    w->status = search_direct(h, w->query, w->n, w->limit, w->out);
    if (w->status != VT_OK) w->error = g_last_error;
  };
  if (members.size() == 1) {
    alone(members[0]);
    return;
  }
  // every query is judged on its own (flat.rs:97-101), as if it had come alone
  std::vector<vt_flat::Waiting *> good;
  {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    const long dim = handle_dim(h);
    for (vt_flat::Waiting *w : members) {
      const int st = h->poisoned ? poisoned_status() : validate_vector(w->query, w->n, dim);
      if (st != VT_OK) {
        w->status = st;
        w->error = st == VT_ERR_POISONED ? g_last_error : std::string();
      } else {
        good.push_back(w);
      }
    }
  }
  if (good.size() < 2) {
    for (vt_flat::Waiting *w : good) alone(w);
    return;
  }
  std::vector<float> qs(good.size() * n);
  for (size_t i = 0; i < good.size(); ++i) std::memcpy(&qs[i * n], good[i]->query, n * sizeof(float));
  std::vector<vt_hits *> outs(good.size(), nullptr);
  const int st = batch_direct(h, qs.data(), good.size(), n, limit, outs.data());
  if (st == VT_OK) {
    for (size_t i = 0; i < good.size(); ++i) *good[i]->out = outs[i];
    return;
  }
  // one query's failure ("metric overflow", a dimension that changed under us) is that query's own
  for (vt_flat::Waiting *w : good) alone(w);
}

int coalesced_search(vt_flat *h, const float *query, size_t n, size_t limit, vt_hits **out) {
  if (limit == 0 || limit > (size_t)vt::kMaxFusedK || n == 0 || !coalescing_enabled()) return search_direct(h, query, n, limit, out);
  vt_flat::Coalescer &co = h->co;
  const unsigned max_active = coalesce_slots(h->approx_bytes.load(std::memory_order_relaxed));
  vt_flat::Waiting me(query, n, limit, out);
  std::vector<vt_flat::Waiting *> members;
  {
    std::unique_lock<std::mutex> lk(co.mu);
    if (co.active < max_active && co.waiting.empty()) {
      co.active += 1;  // nobody to wait for, nobody to take along
    } else {
      co.waiting.push_back(&me);
      co.gather.notify_one();
      me.wake.wait(lk, [&] { return me.state != vt_flat::Waiting::QUEUED; });
      if (me.state == vt_flat::Waiting::DONE) {
        if (me.status != VT_OK) g_last_error = me.error;
        return me.status;
      }
      if (me.state == vt_flat::Waiting::ALONE) {
        lk.unlock();
        return search_direct(h, query, n, limit, out);
      }
      // LEADS (the operation that just finished passed its slot on: `active` already counts this one).
      // Callers that have just been answered are about to come back -- give them a moment (a few
      // % of a pass) before the next pass over the corpus starts without them
      if (co.last_batch > 1 && co.waiting.size() + 1 < co.last_batch) {
        const double window = std::min(300e-6, 0.03 * co.last_seconds);
        const size_t want = co.last_batch - 1;
        co.gather.wait_for(lk, std::chrono::duration<double>(window), [&] { return co.waiting.size() >= want; });
      }
      for (auto it = co.waiting.begin(); it != co.waiting.end() && members.size() + 1 < kCoalesceMax;) {
        if ((*it)->limit == limit && (*it)->n == n) {
          members.push_back(*it);
          it = co.waiting.erase(it);
        } else {
          ++it;
        }
      }
    }
    members.insert(members.begin(), &me);
  }
  // a batch needs strictly current id ranks; a lone search after unsorted inserts does not
  // (lazy ranks): then nobody is made to wait for a re-rank -- everyone searches alone
  bool disband = false;
  if (members.size() > 1 && !h->multi()) {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    disband = shard_stale(h->shards[0].get(), NEED_STRICT_RANKS, limit);
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (disband) {
    {
      std::lock_guard<std::mutex> g(co.mu);
      for (size_t i = 1; i < members.size(); ++i) {
        members[i]->state = vt_flat::Waiting::ALONE;
        members[i]->wake.notify_one();
      }
    }
    members.resize(1);
  }
  try {
    run_coalesced(h, members);
  } catch (...) {  // (host memory, most likely) -- nobody may be left waiting
    for (vt_flat::Waiting *w : members) {
      if (w->out && *w->out) {
        delete *w->out;
        *w->out = nullptr;
      }
      w->status = VT_ERR_NOMEM;
      w->error = "out of host memory";
    }
  }
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  {
    std::lock_guard<std::mutex> g(co.mu);
    // (notified under the lock: a member may return -- and its Waiting leave the stack -- the
    // moment it can take the lock and see DONE)
    for (size_t i = 1; i < members.size(); ++i) {
      members[i]->state = vt_flat::Waiting::DONE;
      members[i]->wake.notify_one();
    }
    co.last_batch = members.size();
    co.last_seconds = seconds;
    if (members.size() > 1) {
      co.batches += 1;
      co.batched_queries += members.size();
    }
    // the longest-waiting caller leads next, in this operation's slot (it takes the others along)
    if (!co.waiting.empty()) {
      co.waiting.front()->state = vt_flat::Waiting::LEADS;
      co.waiting.front()->wake.notify_one();
      co.waiting.pop_front();
    } else {
      co.active -= 1;
    }
  }
  if (me.status != VT_OK) g_last_error = me.error;
  return me.status;
}

}  // namespace

// =============================================================== C ABI
extern "C" {

const char *vt_strerror(int status) {
  switch (status) {
    case VT_OK: return "ok";
    case VT_ERR_EMPTY: return "vector must not be empty";
    case VT_ERR_DIMENSION: return "dimension mismatch";
    case VT_ERR_NON_FINITE: return "vector contains a non-finite value";
    case VT_ERR_OVERFLOW: return "metric overflow";
    case VT_ERR_UNKNOWN_METRIC: return "unknown metric";
    case VT_ERR_PREFIX: return "invalid prefix dimensions";
    case VT_ERR_DIMS_POSITIVE: return "dimensions must be positive";
    case VT_ERR_POISONED: return "flat lock poisoned";
    case VT_ERR_NOMEM: return "out of memory";
    case VT_ERR_DEVICE: return "device error";
    case VT_ERR_UNSUPPORTED: return "unsupported on device";
    case VT_ERR_ARGUMENT: return "bad argument";
    default: return "unknown status";
  }
}

const char *vt_last_error(void) { return g_last_error.c_str(); }
int vt_abi_version(void) { return VT_ABI_VERSION; }

int vt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

size_t vt_hits_len(const vt_hits *h) { return h ? h->ids.size() : 0; }
const char *vt_hits_id(const vt_hits *h, size_t i, size_t *len) {
  *len = h->ids[i].size();
  return h->ids[i].data();
}
float vt_hits_raw(const vt_hits *h, size_t i) { return h->raw[i]; }
uint32_t vt_hits_rank_key(const vt_hits *h, size_t i) { return h->rank_key[i]; }
size_t vt_hits_pack(const vt_hits *h, void *records, size_t cap) {
  if (!h || !records) return 0;
  const size_t n = std::min(cap, h->ids.size());
  unsigned char *out = static_cast<unsigned char *>(records);
  for (size_t i = 0; i < n; ++i, out += VT_HIT_RECORD_BYTES) {
    const uint32_t len = (uint32_t)h->ids[i].size();
    std::memset(out, 0, VT_HIT_RECORD_BYTES);
    std::memcpy(out, &h->rank_key[i], 4);
    std::memcpy(out + 4, &h->raw[i], 4);
    std::memcpy(out + 8, &len, 4);
    std::memcpy(out + 12, h->ids[i].data(), std::min<size_t>(len, VT_HIT_RECORD_ID_BYTES));
  }
  return n;
}
size_t vt_hits_id_bytes(const vt_hits *h) {
  size_t total = 0;
  if (h)
    for (const auto &id : h->ids) total += id.size();
  return total;
}

void vt_hits_export(const vt_hits *h, char *ids, size_t *id_off, float *raw, uint32_t *rank_key) {
  if (!h) return;
  size_t pos = 0;
  for (size_t i = 0; i < h->ids.size(); ++i) {
    if (id_off) id_off[i] = pos;
    if (ids) std::memcpy(ids + pos, h->ids[i].data(), h->ids[i].size());
    pos += h->ids[i].size();
    if (raw) raw[i] = h->raw[i];
    if (rank_key) rank_key[i] = h->rank_key[i];
  }
  if (id_off) id_off[h->ids.size()] = pos;
}

void vt_hits_free(vt_hits *h) { delete h; }

int vt_flat_new_sharded(int metric_code, const int *devices, size_t ndev, vt_flat **out) {
  return guarded([&]() -> int {
  if (!out || !devices || ndev == 0 || ndev > 64) return VT_ERR_ARGUMENT;
  *out = nullptr;
  if (metric_code < VT_L2 || metric_code > VT_JACCARD) return VT_ERR_UNKNOWN_METRIC;
  auto h = std::make_unique<vt_flat>();
  h->metric = metric_code;
  for (size_t s = 0; s < ndev; ++s) {
    auto ix = std::make_unique<Shard>();
    ix->metric = metric_code;
    VT_TRY(ix->ctx.init(devices[s]));
    h->shards.push_back(std::move(ix));
  }
  if (ndev > 1 || std::getenv("VT_SHARD_FORCE_WORKERS")) {
    bool distinct = true;
    for (size_t a = 0; a < ndev; ++a)
      for (size_t b = a + 1; b < ndev; ++b) {
        if (devices[a] == devices[b]) {
          distinct = false;
          continue;
        }
        // bulk loads may hand a shard rows that live on another shard's device
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
          (void)hipSetDevice(devices[a]);
          if (hipDeviceEnablePeerAccess(devices[b], 0) != hipSuccess) (void)hipGetLastError();
        }
        if (hipDeviceCanAccessPeer(&can, devices[b], devices[a]) == hipSuccess && can) {
          (void)hipSetDevice(devices[b]);
          if (hipDeviceEnablePeerAccess(devices[a], 0) != hipSuccess) (void)hipGetLastError();
        }
      }
    for (size_t s = 0; s < ndev; ++s) {
      h->workers.push_back(std::make_unique<Worker>());
      h->workers.back()->start(devices[s]);
    }
    // the shards' lists meet over RCCL when every shard has a device of its own
    // (VT_SHARD_EXCHANGE=host|rccl overrides; vt_flat_set_exchange later)
    const char *want = std::getenv("VT_SHARD_EXCHANGE");
    const bool want_host = want && std::string(want) == "host";
    const bool want_rccl = want && std::string(want) == "rccl";
    if (!want_host && (distinct || want_rccl)) {
      const int st = exchange_setup(h.get());
      if (st == VT_OK) h->exchange = VT_EXCHANGE_RCCL;
      else if (want_rccl) return st;
    }
  }
  *out = h.release();
  return VT_OK;
  });
}

int vt_flat_new(int metric_code, int device, vt_flat **out) { return vt_flat_new_sharded(metric_code, &device, 1, out); }

void vt_flat_free(vt_flat *h) { delete h; }

size_t vt_flat_len(const vt_flat *h) {
  if (!h) return 0;
  std::shared_lock<std::shared_mutex> rl(h->rw);
  return h->total();
}
long vt_flat_dimension(const vt_flat *h) {
  if (!h) return -1;
  std::shared_lock<std::shared_mutex> rl(h->rw);
  return handle_dim(h);
}
int vt_flat_metric(const vt_flat *h) { return h ? h->metric : -1; }
size_t vt_flat_shard_count(const vt_flat *h) { return h ? h->shards.size() : 0; }
int vt_flat_shard_device(const vt_flat *h, size_t shard) {
  return h && shard < h->shards.size() ? h->shards[shard]->ctx.device : -1;
}
size_t vt_flat_shard_len(const vt_flat *h, size_t shard) {
  if (!h || shard >= h->shards.size()) return 0;
  std::shared_lock<std::shared_mutex> rl(h->rw);
  return h->shards[shard]->n;
}
int vt_flat_shard_memory(const vt_flat *h, size_t shard, size_t *row_capacity, size_t *slab_bytes, size_t *slab_chunks) {
  if (!h || shard >= h->shards.size()) return VT_ERR_ARGUMENT;
  std::shared_lock<std::shared_mutex> rl(h->rw);
  const Shard *ix = h->shards[shard].get();
  if (row_capacity) *row_capacity = ix->cap;
  if (slab_bytes) *slab_bytes = ix->slab.bytes;
  if (slab_chunks) *slab_chunks = ix->slab.mapped ? ix->slab.chunks.size() : 0;
  return VT_OK;
}
int vt_flat_coalesce_stats(vt_flat *h, uint64_t *batches, uint64_t *batched_queries) {
  if (!h) return VT_ERR_ARGUMENT;
  std::lock_guard<std::mutex> g(h->co.mu);
  if (batches) *batches = h->co.batches;
  if (batched_queries) *batched_queries = h->co.batched_queries;
  return VT_OK;
}
int vt_flat_route_ids(const vt_flat *h, size_t count, const char *ids, const size_t *id_off, uint32_t *out_shard) {
  if (!h || (count && (!id_off || !out_shard))) return VT_ERR_ARGUMENT;
  const size_t S = h->shards.size();
  for (size_t i = 0; i < count; ++i) out_shard[i] = S > 1 ? shard_of(ids + id_off[i], id_off[i + 1] - id_off[i], S) : 0u;
  return VT_OK;
}
int vt_flat_set_exchange(vt_flat *h, int mode) {
  return guarded([&]() -> int {
  if (!h || (mode != VT_EXCHANGE_HOST && mode != VT_EXCHANGE_RCCL)) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  if (mode == VT_EXCHANGE_RCCL) {
    if (!h->multi()) return fail(VT_ERR_UNSUPPORTED, "a one-shard index has nothing to exchange");
    VT_TRY(exchange_setup(h));
  }
  h->exchange = mode;
  return VT_OK;
  });
}
int vt_flat_exchange(const vt_flat *h) { return h ? h->exchange : -1; }
int vt_flat_rccl_ranks(const vt_flat *h) {
  if (!h || h->comms.empty() || !h->comms[0]) return 0;
  int n = 0;
  if (rccl().CommCount(h->comms[0], &n) != ncclSuccess) return 0;
  return n;
}

int vt_set_default_reduce_order(int order) {
  if (order < VT_ORDER_PAIR || order > VT_ORDER_SSE2) return VT_ERR_ARGUMENT;
  g_default_order = order;
  return VT_OK;
}

int vt_flat_set_reduce_order(vt_flat *h, int order) {
  return guarded([&]() -> int {
  if (!h || order < VT_ORDER_PAIR || order > VT_ORDER_SSE2) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  for (auto &s : h->shards) s->order = order;
  return VT_OK;
  });
}

int vt_flat_insert(vt_flat *h, const char *id, size_t id_len, const float *vector, size_t n) {
  return guarded([&]() -> int {
  if (!h || (!id && id_len) || (!vector && n)) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  if (h->poisoned) return poisoned_status();
  VT_TRY(h->shards[0]->ctx.bind());
  // flat.rs:59-66
  VT_TRY(validate_vector(vector, n, handle_dim(h)));
  const size_t id_off[2] = {0, id_len};
  const size_t val_off[2] = {0, n};
  RowSource src;
  src.host = vector;
  src.off = val_off;
  return store_validated(h, 1, id ? id : "", id_off, src, n);
  });
}

int vt_flat_insert_many(vt_flat *h, size_t count, const char *ids, const size_t *id_off, const float *values,
                        const size_t *value_off) {
  return guarded([&]() -> int {
  if (!h || (count && (!id_off || !value_off))) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  if (h->poisoned) return poisoned_status();
  VT_TRY(h->shards[0]->ctx.bind());
  // flat.rs:69-85: expected = own dimension, else the first row's length;
  // every row is validated before anything is stored.
  long expected = handle_dim(h);
  if (expected < 0 && count > 0) expected = (long)(value_off[1] - value_off[0]);
  for (size_t i = 0; i < count; ++i)
    VT_TRY(validate_vector(values + value_off[i], value_off[i + 1] - value_off[i], expected));
  if (count == 0) return VT_OK;
  RowSource src;
  src.host = values;
  src.off = value_off;
  return store_validated(h, count, ids, id_off, src, (size_t)expected);
  });
}

int vt_flat_load_matrix(vt_flat *h, size_t count, size_t d, const char *ids, const size_t *id_off, const float *rows) {
  return guarded([&]() -> int {
  if (!h || (count && (!id_off || !rows))) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  if (h->poisoned) return poisoned_status();
  VT_TRY(h->shards[0]->ctx.bind());
  long expected = handle_dim(h);
  if (expected < 0 && count > 0) expected = (long)d;
  VT_TRY(validate_matrix(rows, count, d, expected));
  if (count == 0) return VT_OK;
  RowSource src;
  src.host = rows;
  src.d = d;
  return store_validated(h, count, ids, id_off, src, d);
  });
}

int vt_flat_load_device_matrix(vt_flat *h, size_t count, size_t d, const char *ids, const size_t *id_off,
                               const void *device_rows) {
  return guarded([&]() -> int {
  if (!h || (count && (!id_off || !device_rows))) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  if (h->poisoned) return poisoned_status();
  if (count == 0) return VT_OK;
  if (count > 0xFFFFFFF0ull) return fail(VT_ERR_UNSUPPORTED, "more than 2^32-16 rows");
  const long expected = handle_dim(h) < 0 ? (long)d : handle_dim(h);
  if (d == 0) return VT_ERR_EMPTY;
  if ((long)d != expected) return VT_ERR_DIMENSION;
  const float *rows = static_cast<const float *>(device_rows);
  // finiteness is checked where the rows live (a shard on that device, else shard 0 over the peer mapping)
  const int home = device_of_pointer(device_rows);
  Shard *checker = h->shards[0].get();
  for (auto &s : h->shards)
    if (s->ctx.device == home) {
      checker = s.get();
      break;
    }
  Ctx &c = checker->ctx;
  VT_TRY(c.bind());
  VT_HIP(hipDeviceSynchronize());  // the producer may have used another stream
  int non_finite = 0;
  VT_HIP(hipMemsetAsync(c.dFlag.p, 0, sizeof(int), c.stream));
  VT_HIP(vt::launch_check_finite(rows, d, (uint32_t)count, (uint32_t)d, c.dFlag.p, c.stream));
  VT_HIP(hipMemcpyAsync(&non_finite, c.dFlag.p, sizeof(int), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  if (non_finite != 0) return VT_ERR_NON_FINITE;
  VT_TRY(h->shards[0]->ctx.bind());
  RowSource src;
  src.device = rows;
  src.d = d;
  return store_validated(h, count, ids, id_off, src, d);
  });
}

int vt_flat_delete(vt_flat *h, const char *id, size_t id_len) {
  return guarded([&]() -> int {
  if (!h || (!id && id_len)) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  if (h->poisoned) return poisoned_status();
  bool began = false;
  int st;
  if (!h->multi()) {
    Shard *ix = h->shards[0].get();
    VT_TRY(ix->ctx.bind());
    st = shard_delete(ix, id, id_len, &began);
  } else {
    const size_t s = shard_of(id ? id : "", id_len, h->shards.size());
    st = on_shards(h, std::vector<size_t>{s}, [&](size_t t) -> int { return shard_delete(h->shards[t].get(), id, id_len, &began); });
    if (h->total() == 0) h->dim = -1;  // flat.rs:90-92: an emptied index forgets its dimension
  }
  if (st != VT_OK && began) h->poisoned = true;
  refresh_approx_bytes(h);
  return st;
  });
}

int vt_flat_search(vt_flat *h, const float *query, size_t n, size_t limit, vt_hits **out) {
  return guarded([&]() -> int {
  if (!h || !out || (!query && n)) return VT_ERR_ARGUMENT;
  *out = nullptr;
  return coalesced_search(h, query, n, limit, out);
  });
}

int vt_rank_ids(const char *ids, const size_t *id_off, size_t count, uint32_t *out_rank) {
  return guarded([&]() -> int {
  if (count && (!id_off || !out_rank)) return VT_ERR_ARGUMENT;
  if (count > 0xFFFFFFF0ull) return fail(VT_ERR_UNSUPPORTED, "more than 2^32-16 ids");
  std::vector<uint32_t> order(count);
  for (size_t i = 0; i < count; ++i) order[i] = (uint32_t)i;
  auto less = [&](uint32_t a, uint32_t b) {
    const size_t la = id_off[a + 1] - id_off[a], lb = id_off[b + 1] - id_off[b];
    const size_t m = std::min(la, lb);
    const int c = m ? std::memcmp(ids + id_off[a], ids + id_off[b], m) : 0;
    if (c) return c < 0;
    if (la != lb) return la < lb;
    return a < b;  // equal ids: input order
  };
  parallel_sort(order, less);
  for (size_t i = 0; i < count; ++i) out_rank[order[i]] = (uint32_t)i;
  return VT_OK;
  });
}

// ---- process-per-GPU sharding (vettore_amd/sharded.py): the caller owns the collective.
// One-shard handles only; these calls run on the primary context under the exclusive lock.
#define VT_SINGLE_SHARD_ONLY(h)                                                                                     \
  if ((h)->multi()) return fail(VT_ERR_UNSUPPORTED, "a multi-shard handle runs its own exchange (vt_flat_search)")

int vt_flat_set_id_ranks(vt_flat *h, const uint32_t *ranks, size_t count) {
  return guarded([&]() -> int {
  if (!h || (count && !ranks)) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  if (h->poisoned) return poisoned_status();
  VT_SINGLE_SHARD_ONLY(h);
  Shard *ix = h->shards[0].get();
  VT_TRY(ix->ctx.bind());
  if (count != ix->n) return VT_ERR_DIMENSION;
  ix->rank_host.assign(ranks, ranks + count);
  ix->unranked = 0;
  ix->ranks_clean = true;
  ix->external_ranks = true;
  ix->external_expected = true;
  ix->external_epoch = ix->epoch;
  ix->max_rank = kUnranked - 1;  // an appended id can no longer extend the ranks in place
  return index_sync_ranks(ix, true);
  });
}

void *vt_flat_stream(vt_flat *h) { return h && !h->multi() ? static_cast<void *>(h->shards[0]->ctx.stream) : nullptr; }

int vt_flat_search_begin(vt_flat *h, const float *query, size_t n, size_t limit, void *device_block) {
  return guarded([&]() -> int {
  if (!h || !device_block || (!query && n)) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  if (h->poisoned) return poisoned_status();
  VT_SINGLE_SHARD_ONLY(h);
  Shard *ix = h->shards[0].get();
  Ctx &c = ix->ctx;
  VT_TRY(c.bind());
  if (ix->external_expected && (!ix->external_ranks || ix->epoch != ix->external_epoch)) {
    // The ranks installed by vt_flat_set_id_ranks described another row set: keys of this
    // shard no longer compare with the other shards', and rows have moved under the caller's
    // id table.  The block says so (the merge hands the bit to every rank), nothing is scanned.
    if (limit == 0 || limit > (size_t)vt::kMaxFusedK) return fail(VT_ERR_UNSUPPORTED, "search_begin needs 1 <= limit <= 256");
    VT_TRY(validate_vector(query, n, ix->dim));
    VT_HIP(hipMemsetAsync(device_block, 0, 16, c.stream));
    VT_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(device_block), kStatusStaleRanks, 1, c.stream));
    c.begin_rows = 0;
    return VT_OK;
  }
  VT_TRY(shard_prepare(ix, NEED_STRICT_RANKS, limit));
  return shard_begin(ix, c, query, n, limit, device_block);  // nothing waited for: the caller's collective queues behind
  });
}

int vt_flat_merge_gathered(vt_flat *h, const void *device_blocks, size_t world, size_t limit, size_t block_bytes,
                           uint64_t *keys, uint32_t *rows, float *raw, uint32_t *shard, size_t *count) {
  return guarded([&]() -> int {
  if (!h || !device_blocks || !keys || !rows || !raw || !shard || !count) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  if (h->poisoned) return poisoned_status();
  VT_SINGLE_SHARD_ONLY(h);
  Ctx &c = h->shards[0]->ctx;
  VT_TRY(c.bind());
  if (limit == 0 || limit > (size_t)vt::kMaxFusedK || world == 0) return VT_ERR_ARGUMENT;
  VT_HIP(vt::launch_merge_blocks(device_blocks, (uint32_t)world, (uint32_t)limit, (uint32_t)block_bytes, c.dResMapped,
                                 c.dShardMapped, c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  VT_TRY(settle_begin_profile(c));
  if (c.hRes.p->status & kStatusStaleRanks)
    return fail(VT_ERR_UNSUPPORTED, "stale id ranks: a shard was mutated after vt_flat_set_id_ranks");
  if (c.hRes.p->status == VT_ERR_OVERFLOW) return VT_ERR_OVERFLOW;
  const uint32_t got = c.hRes.p->count;
  for (uint32_t i = 0; i < got; ++i) {
    keys[i] = c.hRes.p->e[i].key;
    rows[i] = c.hRes.p->e[i].row;
    raw[i] = c.hRes.p->e[i].raw;
    shard[i] = c.hShard.p[i];
  }
  *count = got;
  return VT_OK;
  });
}

int vt_flat_search_batch(vt_flat *h, const float *queries, size_t nq, size_t d, size_t limit, vt_hits **out) {
  return guarded([&]() -> int {
  if (!h || !out || (nq && !queries && d)) return VT_ERR_ARGUMENT;
  return batch_direct(h, queries, nq, d, limit, out);
  });
}

int vt_flat_quantized_search(vt_flat *h, const float *query, size_t n, size_t candidates, size_t limit, vt_hits **out) {
  return guarded([&]() -> int {
  if (!h || !out || (!query && n)) return VT_ERR_ARGUMENT;
  *out = nullptr;
  if (h->multi()) {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    if (h->poisoned) return poisoned_status();
    return quantized_multi(h, query, n, candidates, limit, out);
  }
  return read_single(h, NEED_STRICT_RANKS | NEED_BITS, limit, [&](Shard *ix, Ctx &c) -> int {
    return quantized_ready(ix, c, query, n, candidates, limit, out);
  });
  });
}

int vt_flat_funnel_search(vt_flat *h, const float *query, size_t n, const size_t *stages, size_t nstages,
                          size_t candidates, size_t limit, vt_hits **out) {
  return guarded([&]() -> int {
  if (!h || !out || (!query && n) || (nstages && !stages)) return VT_ERR_ARGUMENT;
  *out = nullptr;
  if (h->multi()) {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    if (h->poisoned) return poisoned_status();
    return funnel_multi(h, query, n, stages, nstages, candidates, limit, out);
  }
  return read_single(h, NEED_STRICT_RANKS, limit, [&](Shard *ix, Ctx &c) -> int {
    return funnel_ready(ix, c, query, n, stages, nstages, candidates, limit, out);
  });
  });
}

int vt_flat_hybrid_search(vt_flat *h, const float *query, size_t n, const int *kinds, const size_t *candidates,
                          const size_t *stage_off, const size_t *stages, size_t ngen, size_t limit, vt_hits **out) {
  return guarded([&]() -> int {
  if (!h || !out || (!query && n) || (ngen && (!kinds || !candidates || !stage_off))) return VT_ERR_ARGUMENT;
  *out = nullptr;
  if (h->multi()) {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    if (h->poisoned) return poisoned_status();
    return hybrid_multi(h, query, n, kinds, candidates, stage_off, stages, ngen, limit, out);
  }
  unsigned need = NEED_STRICT_RANKS;
  for (size_t i = 0; i < ngen; ++i)
    if (kinds[i] == VT_GEN_QUANTIZED) need |= NEED_BITS;
  return read_single(h, need, limit, [&](Shard *ix, Ctx &c) -> int {
    return hybrid_ready(ix, c, query, n, kinds, candidates, stage_off, stages, ngen, limit, out);
  });
  });
}

int vt_vector_top_k(int device, size_t count, const char *ids, const size_t *id_off, const float *values,
                    const size_t *value_off, const float *query, size_t nq, int metric_code, size_t dimensions,
                    size_t limit, vt_hits **out) {
  return guarded([&]() -> int {
  if (!out || (count && (!id_off || !value_off))) return VT_ERR_ARGUMENT;
  *out = nullptr;
  // nifs.rs:158-161: metric decode first, then search.rs:38-73
  if (metric_code < VT_L2 || metric_code > VT_JACCARD) return VT_ERR_UNKNOWN_METRIC;
  if (dimensions == 0 || dimensions > nq) return VT_ERR_PREFIX;
  VT_TRY(validate_finite(query, dimensions));
  // the reference walks the batch in order and stops at the first error;
  // rows before the first invalid one may still overflow and win the race
  size_t good = count;
  int first_error = VT_OK;
  for (size_t i = 0; i < count; ++i) {
    const size_t len = value_off[i + 1] - value_off[i];
    int e = VT_OK;
    if (dimensions > len) e = VT_ERR_DIMENSION;
    else e = validate_finite(values + value_off[i], dimensions);
    if (e != VT_OK) {
      good = i;
      first_error = e;
      break;
    }
  }
  if (dimensions > 0x7fffffffu || vt::scan_lds_bytes((uint32_t)dimensions, 1) == 0)
    return fail(VT_ERR_UNSUPPORTED, "prefix dimension exceeds what the scan kernel stages in LDS");
  if (count > 0xFFFFFFF0ull) return fail(VT_ERR_UNSUPPORTED, "more than 2^32-16 rows");
  Ctx *cp = nullptr;
  VT_TRY(stateless_ctx(device, &cp));
  Ctx &c = *cp;
  std::lock_guard<std::mutex> g(g_ctx_mu);
  std::vector<vt::Entry> entries;
  if (good > 0) {
    const uint32_t d = (uint32_t)dimensions, ld = vt::padded_dim(d);
    const uint32_t n = (uint32_t)good;
    const uint32_t cap = round_up_u32(n, vt::kTileRows);
    std::vector<float> packed((size_t)cap * ld, 0.0f);
    for (size_t i = 0; i < good; ++i) std::memcpy(&packed[i * ld], values + value_off[i], (size_t)d * sizeof(float));
    std::vector<uint32_t> rank;
    ranks_for_ids(ids, id_off, good, rank);
    DevBuf<float> dX;
    DevBuf<uint32_t> dRank;
    VT_TRY(dX.ensure(packed.size()));
    VT_TRY(dRank.ensure(n));
    VT_HIP(hipMemcpyAsync(dX.p, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, c.stream));
    VT_HIP(hipMemcpyAsync(dRank.p, rank.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
    uint32_t qnz = 0;
    VT_TRY(upload_query(c, query, dimensions, &qnz));
    const size_t want = first_error == VT_OK ? limit : (size_t)1;  // only the overflow flag matters then
    if (metric_code == VT_COSINE) {
      VT_TRY(c.dCandKeys.ensure(n));
      VT_TRY(c.dCandPay.ensure(n));
      vt::CosineRerankArgs a{};
      a.X = dX.p;
      a.stride = ld;
      a.q = c.dQ.p;
      a.id_rank = dRank.p;
      a.gather = nullptr;
      a.gather_stride = 0;
      a.n = n;
      a.d = d;
      a.out_keys = c.dCandKeys.p;
      a.out_pay = c.dCandPay.p;
      a.status = c.dStatus.p;
      VT_HIP(vt::launch_cosine_rerank(a, c.stream));
      // limit == 0 still has to surface "metric overflow": select one
      VT_TRY(collect_from_keys(c, c.dCandKeys.p, c.dCandPay.p, n, std::max<size_t>(want, 1), entries));
      if (limit == 0) entries.clear();
    } else {
      ScanJob j{};
      j.X = dX.p;
      j.stride = ld;
      j.id_rank = dRank.p;
      j.n = n;
      j.d = d;
      j.metric = metric_code;
      j.order = g_default_order;
      j.q_nonzero = qnz;
      // limit == 0 still has to surface "metric overflow": scan for one hit
      VT_TRY(run_scan(c, j, std::max<size_t>(want, 1), entries, false));
      if (limit == 0) entries.clear();
    }
  }
  if (first_error != VT_OK) return first_error;
  return hits_from_batch(ids, id_off, entries, out);
  });
}

int vt_binary_top_k(int device, size_t count, const char *ids, const size_t *id_off, const uint64_t *words,
                    const size_t *word_off, const uint64_t *query, size_t nq, size_t dimensions, size_t limit,
                    vt_hits **out) {
  return guarded([&]() -> int {
  if (!out || (count && (!id_off || !word_off))) return VT_ERR_ARGUMENT;
  *out = nullptr;
  // search.rs:82-84: the query is validated against itself first
  const size_t W = (dimensions + 63) / 64;
  if (dimensions == 0) return VT_ERR_DIMS_POSITIVE;
  if (nq != W) return VT_ERR_DIMENSION;
  for (size_t i = 0; i < count; ++i)
    if (word_off[i + 1] - word_off[i] != W) return VT_ERR_DIMENSION;
  if (count == 0 || limit == 0) return empty_hits(out);
  if (count > 0xFFFFFFF0ull || dimensions > 0x7fffffffu) return fail(VT_ERR_UNSUPPORTED, "batch too large");
  Ctx *cp = nullptr;
  VT_TRY(stateless_ctx(device, &cp));
  Ctx &c = *cp;
  std::lock_guard<std::mutex> g(g_ctx_mu);
  const uint32_t n = (uint32_t)count;
  // K4 reads the tiled layout: [tile of 64 rows][word pair][row][2]
  const uint32_t pairs = (uint32_t)((W + 1) / 2);
  std::vector<uint64_t> packed(vt::hamming_matrix_words(n, (uint32_t)W), 0);
  for (size_t i = 0; i < count; ++i)
    for (size_t w = 0; w < W; ++w) packed[vt::hamming_word_index((uint32_t)i, (uint32_t)w, pairs)] = words[word_off[i] + w];
  std::vector<uint32_t> rank;
  ranks_for_ids(ids, id_off, count, rank);
  DevBuf<uint64_t> dBits, dQ;
  DevBuf<uint32_t> dRank;
  VT_TRY(dBits.ensure(packed.size()));
  VT_TRY(dQ.ensure(W));
  VT_TRY(dRank.ensure(n));
  VT_HIP(hipMemcpyAsync(dBits.p, packed.data(), packed.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c.stream));
  VT_HIP(hipMemcpyAsync(dQ.p, query, W * sizeof(uint64_t), hipMemcpyHostToDevice, c.stream));
  VT_HIP(hipMemcpyAsync(dRank.p, rank.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
  std::vector<vt::Entry> entries;
  VT_TRY(run_hamming(c, dBits.p, dQ.p, dRank.p, n, (uint32_t)dimensions, limit, entries, false));
  return hits_from_batch(ids, id_off, entries, out);
  });
}

int vt_normalize_l2(int device, size_t count, size_t d, const float *in, float *out) {
  return guarded([&]() -> int {
  if ((count && d) && (!in || !out)) return VT_ERR_ARGUMENT;
  // distances.rs:350-361: finiteness first
  VT_TRY(validate_finite(in, count * d));
  if (count == 0 || d == 0) return VT_OK;
  if (count > 0xFFFFFFF0ull || d > 0x7fffffffu) return fail(VT_ERR_UNSUPPORTED, "batch too large");
  Ctx *cp = nullptr;
  VT_TRY(stateless_ctx(device, &cp));
  Ctx &c = *cp;
  std::lock_guard<std::mutex> g(g_ctx_mu);
  DevBuf<float> dIn, dOut;
  VT_TRY(dIn.ensure(count * d));
  VT_TRY(dOut.ensure(count * d));
  VT_HIP(hipMemcpyAsync(dIn.p, in, count * d * sizeof(float), hipMemcpyHostToDevice, c.stream));
  VT_HIP(vt::launch_normalize_l2(dIn.p, (uint32_t)count, (uint32_t)d, dOut.p, c.stream));
  VT_HIP(hipMemcpyAsync(out, dOut.p, count * d * sizeof(float), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  return VT_OK;
  });
}

int vt_compress_sign_bits(int device, size_t count, size_t d, const float *in, uint64_t *out) {
  return guarded([&]() -> int {
  if ((count && d) && (!in || !out)) return VT_ERR_ARGUMENT;
  if (count == 0 || d == 0) return VT_OK;
  if (count > 0xFFFFFFF0ull || d > 0x7fffffffu) return fail(VT_ERR_UNSUPPORTED, "batch too large");
  Ctx *cp = nullptr;
  VT_TRY(stateless_ctx(device, &cp));
  Ctx &c = *cp;
  std::lock_guard<std::mutex> g(g_ctx_mu);
  const size_t W = (d + 63) / 64;
  DevBuf<float> dIn;
  DevBuf<uint64_t> dOut;
  VT_TRY(dIn.ensure(count * d));
  VT_TRY(dOut.ensure(count * W));
  VT_HIP(hipMemcpyAsync(dIn.p, in, count * d * sizeof(float), hipMemcpyHostToDevice, c.stream));
  VT_HIP(vt::launch_sign_pack(dIn.p, d, (uint32_t)count, (uint32_t)d, dOut.p, 0, c.stream));
  VT_HIP(hipMemcpyAsync(out, dOut.p, count * W * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream));
  VT_HIP(hipStreamSynchronize(c.stream));
  return VT_OK;
  });
}

int vt_flat_set_profiling(vt_flat *h, int enabled) {
  return guarded([&]() -> int {
  if (!h) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  for (auto &s : h->shards) s->for_each_ctx([&](Ctx &c) { c.profiling = enabled != 0; });
  return VT_OK;
  });
}

// Sums over every shard and every reader context of the handle.
int vt_flat_get_profile(vt_flat *h, vt_profile *out, int reset) {
  return guarded([&]() -> int {
  if (!h || !out) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  vt_profile t = h->xprof;
  for (auto &s : h->shards)
    s->for_each_ctx([&](Ctx &c) {
      const vt_profile &p = c.prof;
      t.scan_launches += p.scan_launches;
      t.scan_ms += p.scan_ms;
      t.scan_rows += p.scan_rows;
      t.scan_bytes += p.scan_bytes;
      t.hamming_launches += p.hamming_launches;
      t.hamming_ms += p.hamming_ms;
      t.hamming_bytes += p.hamming_bytes;
      t.merge_launches += p.merge_launches;
      t.merge_ms += p.merge_ms;
      t.batch_launches += p.batch_launches;
      t.batch_ms += p.batch_ms;
      t.batch_flops += p.batch_flops;
      t.batch_queries += p.batch_queries;
      t.batch_fallbacks += p.batch_fallbacks;
      t.prefix_launches += p.prefix_launches;
      t.prefix_ms += p.prefix_ms;
      t.prefix_bytes += p.prefix_bytes;
      if (reset) c.prof = vt_profile{};
    });
  if (reset) h->xprof = vt_profile{};
  *out = t;
  return VT_OK;
  });
}

}  // extern "C"
