// vt_base.h -- statuses, validation, thread helpers, device / pinned buffers and the row slab.
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

thread_local std::string g_last_error;
// Lane order of wide::f32x8::reduce_add assumed for new indexes (include/vettore_flat.h;
// DESIGN.md "summation order" says why SSE2 and how to pin it): VT_REDUCE_ORDER overrides.
// (every VT_* variable is read once, when the library is loaded: csrc/vt_env.h)
static_assert(VT_ORDER_PAIR == 0 && VT_ORDER_AVX == 1 && VT_ORDER_SEQ == 2 && VT_ORDER_SSE2 == 3, "vt_env.h parses VT_REDUCE_ORDER to these");
inline int default_order() { return (int)vt::env::get(vt::env::REDUCE_ORDER); }  // (vt_set_default_reduce_order stores there)

// Which matrix-core pass nominates a batch's candidates: K2b (operands
// rounded to bf16, HBM-bound) unless VT_BATCH_NOMINATE=f32 asks for K2.  Results are the same
// bit for bit either way -- the exact kernel decides.
static_assert(VT_NOMINATE_F32 == 1 && VT_NOMINATE_BF16 == 2, "vt_env.h parses VT_BATCH_NOMINATE to these");
inline int default_nominate() { return (int)vt::env::get(vt::env::BATCH_NOMINATE); }
// Whether new indexes keep a bf16 shadow of their rows for that pass (include/vettore_flat.h,
// vt_flat_set_batch_shadow): VT_BATCH_SHADOW=0 says no.
static_assert(VT_SHADOW_OFF == 0 && VT_SHADOW_AUTO == 1, "vt_env.h parses VT_BATCH_SHADOW to these");
inline int default_shadow() { return (int)vt::env::get(vt::env::BATCH_SHADOW); }
inline int default_single_nominate() { return vt::env::get(vt::env::SINGLE_NOMINATE) == 1 ? 1 : 0; }
// smallest sample rank K2b's threshold is taken from (VT_BF16_MIN_RANK: tools/nominate_probe.py sweeps it).
// The count of rows passing a threshold taken at sample rank r is Gamma(r)-distributed around its
// mean: at r = 4 one query in ~700 drew a threshold so high that its k-th hit could not clear it by
// the margin (one 5-ms single scan per three or four 256-query batches at 10 M x 768); at r = 6 none
// did in 15 000 queries, for +50 % candidates (+0.07 ms in the scoring pass's append path, +0.08 ms
// of re-scoring).  Per 256-query batch, measured: r = 4: 6.20 ms, 6: 6.09, 8: 6.23, 12: 6.39.
uint32_t bf16_min_rank() {
  const long v = vt::env::get(vt::env::BF16_MIN_RANK);
  return v >= 1 && v <= 4096 ? (uint32_t)v : 6u;
}

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

// A mutation's body as a status: whatever it throws (an allocation failing half way through
// the id table's change, say) comes back like any other failure, so that the caller's
// "failed after it began => poisoned" rule sees it (an exception used to unwind
// past that rule and leave a half-updated index usable).
template <typename F>
int no_throw(F &&f) noexcept {
  try {
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
// Finiteness on the bit patterns (auto-vectorises: the bulk loads run it over tens of GB, a batch of 256 queries over
// 200 000 floats -- the scalar isfinite loop with its early exit took 0.1-0.2 ms of a 4.3-ms batch).
inline bool all_finite_bits(const float *v, size_t n) {
  const uint32_t *u = reinterpret_cast<const uint32_t *>(v);
  uint32_t worst = 0;
  for (size_t i = 0; i < n; ++i) worst = std::max(worst, u[i] & 0x7f800000u);
  return worst != 0x7f800000u;
}

int validate_vector(const float *v, size_t n, long dimension) {
  if (n == 0) return VT_ERR_EMPTY;
  if (dimension >= 0 && n != (size_t)dimension) return VT_ERR_DIMENSION;
  return all_finite_bits(v, n) ? VT_OK : VT_ERR_NON_FINITE;
}

int validate_finite(const float *v, size_t n) { return all_finite_bits(v, n) ? VT_OK : VT_ERR_NON_FINITE; }

// Splits [0, n) over up to 16 host threads (bulk ingest: validation and staging copies
// are plain memory passes).  `f(lo, hi)` must not throw.
template <class F>
void parallel_for(size_t n, size_t grain, F f, unsigned max_threads = 16u) {
  unsigned threads = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), max_threads);
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
template <typename T, typename Less>
void parallel_sort(std::vector<T> &idx, Less less) {
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
  T *dev = nullptr;  // the same block as a kernel sees it (mapped()), while it is the same block
  ~PinnedBuf() { release(); }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    dev = nullptr;
    count = 0;
  }
  // the device-side address of the (host-mapped) block: kernels write results straight into it
  T *mapped() {
    if (!dev && p && hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), p, 0) != hipSuccess) {
      (void)hipGetLastError();
      dev = nullptr;
    }
    return dev;
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
    const long mb = vt::env::get(vt::env::SLAB_CHUNK_MB);  // (tests: small chunks, so that small corpora cross chunk borders)
    return mb > 0 ? (size_t)mb << 20 : (size_t)1 << 30;
  }
  static bool mapping_allowed() {
    return vt::env::get(vt::env::SLAB) != 1;  // (VT_SLAB=malloc)
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

}  // namespace
