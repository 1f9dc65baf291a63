// vt_index.cpp -- host side of libvettore_hip.so: the C ABI of
// include/vettore_flat.h, the id table, and the orchestration of the gfx950
// kernels in vt_device.hip.  It mirrors native/vettore/src/nifs.rs (boundary),
// flat.rs (index semantics) and search.rs (stateless helpers) of the reference;
// all metric arithmetic and all selection run on the GPU.
#include "../../include/vettore_flat.h"
#include "vt_device.h"
#define VT_ENV_IMPLEMENTATION  // the library's one copy of the settings table (filled when the library is loaded)
#include "vt_env.h"

#include <algorithm>
#include <atomic>
#include <chrono>
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

// The host side in reading order (one translation unit; see the note at the top of each part):
#include "host/vt_idtable.h"
#include "host/vt_concurrency.h"
#include "host/vt_base.h"
#include "host/vt_types.h"
#include "host/vt_select.h"
#include "host/vt_store.h"
#include "host/vt_search.h"
#include "host/vt_batch.h"
#include "host/vt_quantized.h"
#include "host/vt_funnel.h"
#include "host/vt_hybrid.h"
#include "host/vt_multi.h"
#include "host/vt_coalesce.h"

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

int vt_debug_set(const char *name, long value) {
  const int k = vt::env::find(name);
  // (a value the setting's own parser could not have produced is refused: reduce_order = 7 would hand new indexes a lane
  // order no kernel has -- ADVICE r5)
  if (k < 0 || !vt::env::valid(k, value)) return VT_ERR_ARGUMENT;
  vt::env::set((vt::env::Key)k, value);
  return VT_OK;
}
int vt_debug_get(const char *name, long *value) {
  const int k = vt::env::find(name);
  if (k < 0 || !value) return VT_ERR_ARGUMENT;
  *value = vt::env::load((vt::env::Key)k);
  return VT_OK;
}

int vt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int vt_device_read_peak(int device, size_t bytes, int reps, double *gbps) {
  return guarded([&]() -> int {
  if (!gbps || bytes < (1u << 20) || reps < 1) return VT_ERR_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(VT_ERR_DEVICE, "no HIP device visible: libvettore_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(VT_ERR_DEVICE, "device ordinal out of range");
  VT_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  VT_HIP(hipGetDeviceProperties(&prop, device));
  bytes &= ~(size_t)15;
  void *buf = nullptr;
  float *sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int st = VT_OK;
  double best = 0.0;
  auto body = [&]() -> int {
    VT_HIP(hipMalloc(&buf, bytes));
    VT_HIP(hipMalloc(reinterpret_cast<void **>(&sink), 16));
    VT_HIP(vt::launch_peak_fill(buf, bytes, nullptr));
    VT_HIP(hipEventCreate(&e0));
    VT_HIP(hipEventCreate(&e1));
    const uint32_t blocks = (uint32_t)std::max(1, prop.multiProcessorCount);  // one 8-wave block per CU (80 KiB of LDS each)
    const size_t read_bytes = vt::read_peak_bytes(bytes);
    for (int r = 0; r <= reps; ++r) {  // pass 0 warms up
      VT_HIP(hipEventRecord(e0, nullptr));
      VT_HIP(vt::launch_read_peak(buf, bytes, sink, blocks, nullptr));
      VT_HIP(hipEventRecord(e1, nullptr));
      VT_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      VT_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms > 0.f) best = std::max(best, (double)read_bytes / (ms * 1e-3) / 1e9);
    }
    return VT_OK;
  };
  st = body();
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (sink) (void)hipFree(sink);
  if (buf) (void)hipFree(buf);
  if (st == VT_OK) *gbps = best;
  return st;
  });
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
void vt_hits_pack_many(const vt_hits *const *hits, size_t nq, size_t limit, void *blocks) {
  if (!hits || !blocks) return;
  unsigned char *out = static_cast<unsigned char *>(blocks);
  const size_t block = (limit + 1) * VT_HIT_RECORD_BYTES;
  for (size_t i = 0; i < nq; ++i) {
    unsigned char *b = out + i * block;
    std::memset(b, 0, VT_HIT_RECORD_BYTES);
    const uint32_t n = hits[i] ? (uint32_t)vt_hits_pack(hits[i], b + VT_HIT_RECORD_BYTES, limit) : 0u;
    uint32_t long_ids = 0;
    for (uint32_t j = 0; j < n; ++j) long_ids |= hits[i]->ids[j].size() > VT_HIT_RECORD_ID_BYTES ? 1u : 0u;
    std::memcpy(b, &n, 4);
    std::memcpy(b + 4, &long_ids, 4);
  }
}

int vt_hit_blocks_merge(const void *blocks, size_t world, size_t nq, size_t limit, void *out_blocks) {
  return guarded([&]() -> int {
  if (!blocks || !out_blocks || world == 0) return VT_ERR_ARGUMENT;
  const unsigned char *in = static_cast<const unsigned char *>(blocks);
  unsigned char *out = static_cast<unsigned char *>(out_blocks);
  const size_t block = (limit + 1) * VT_HIT_RECORD_BYTES;
  struct Rec {
    uint32_t key;
    const unsigned char *p;
  };
  auto less = [](const Rec &a, const Rec &b) {
    if (a.key != b.key) return a.key < b.key;
    uint32_t la, lb;
    std::memcpy(&la, a.p + 8, 4);
    std::memcpy(&lb, b.p + 8, 4);
    const size_t ia = std::min<size_t>(la, VT_HIT_RECORD_ID_BYTES), ib = std::min<size_t>(lb, VT_HIT_RECORD_ID_BYTES);
    const int c = std::memcmp(a.p + 12, b.p + 12, std::min(ia, ib));
    if (c) return c < 0;
    return la < lb;  // bytewise String order: a prefix sorts first
  };
  parallel_for(nq, 32, [&](size_t lo, size_t hi) {
    std::vector<Rec> recs;
    for (size_t q = lo; q < hi; ++q) {
      recs.clear();
      uint32_t long_ids = 0;
      for (size_t r = 0; r < world; ++r) {
        const unsigned char *b = in + (r * nq + q) * block;
        uint32_t n, fl;
        std::memcpy(&n, b, 4);
        std::memcpy(&fl, b + 4, 4);
        long_ids |= fl;
        n = (uint32_t)std::min<size_t>(n, limit);
        for (uint32_t j = 0; j < n; ++j) {
          const unsigned char *p = b + (size_t)(j + 1) * VT_HIT_RECORD_BYTES;
          Rec rc;
          std::memcpy(&rc.key, p, 4);
          rc.p = p;
          recs.push_back(rc);
        }
      }
      const size_t k = std::min(limit, recs.size());
      std::partial_sort(recs.begin(), recs.begin() + k, recs.end(), less);
      unsigned char *o = out + q * block;
      std::memset(o, 0, block);
      const uint32_t n = (uint32_t)k;
      std::memcpy(o, &n, 4);
      std::memcpy(o + 4, &long_ids, 4);
      for (size_t j = 0; j < k; ++j) std::memcpy(o + (j + 1) * VT_HIT_RECORD_BYTES, recs[j].p, VT_HIT_RECORD_BYTES);
    }
  });
  return VT_OK;
  });
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
void vt_hits_free_many(vt_hits **hs, size_t n) {
  if (!hs) return;
  for (size_t i = 0; i < n; ++i) {
    delete hs[i];
    hs[i] = nullptr;
  }
}

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
  if (ndev > 1 || vt::env::on(vt::env::SHARD_FORCE_WORKERS)) {
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
    const long want = vt::env::get(vt::env::SHARD_EXCHANGE);  // (VT_SHARD_EXCHANGE=host|rccl)
    const bool want_host = want == 1;
    const bool want_rccl = want == 2;
    std::string note = std::to_string(ndev) + " shards: ";
    if (want_host) {
      note += "host exchange (VT_SHARD_EXCHANGE=host)";
    } else if (!distinct && !want_rccl) {
      note += "host exchange (several shards share a device: RCCL wants one rank per device)";
    } else {
      const int st = exchange_setup(h.get());
      if (st == VT_OK) {
        h->exchange = VT_EXCHANGE_RCCL;
        note += "RCCL all-gather of the per-shard top-k lists, " + std::to_string(vt_flat_rccl_ranks(h.get())) +
                " ranks (ncclCommInitAll in this process); limits above 256, batches and staged searches take host-mapped lists";
      } else if (want_rccl) {
        return st;
      } else {
        note += "host exchange (RCCL refused: " + g_last_error + ")";
      }
    }
    h->exchange_note = note;
    if (vt::env::on(vt::env::LOG)) std::fprintf(stderr, "[vt] %s\n", note.c_str());
  }
  *out = h.release();
  return VT_OK;
  });
}

int vt_flat_new(int metric_code, int device, vt_flat **out) { return vt_flat_new_sharded(metric_code, &device, 1, out); }

void vt_flat_free(vt_flat *h) {
  if (h && h->wedged.load()) {
    // (see vt_flat::wedged: freeing would wait for a collective that never completes; the workers are detached with
    // the handle, the memory goes when the process does)
    return;
  }
  delete h;
}

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
const char *vt_flat_exchange_note(const vt_flat *h) { return h ? h->exchange_note.c_str() : ""; }
int vt_flat_rccl_ranks(const vt_flat *h) {
  if (!h || h->comms.empty() || !h->comms[0]) return 0;
  int n = 0;
  if (rccl().CommCount(h->comms[0], &n) != ncclSuccess) return 0;
  return n;
}

int vt_set_default_reduce_order(int order) {
  if (order < VT_ORDER_PAIR || order > VT_ORDER_SSE2) return VT_ERR_ARGUMENT;
  vt::env::set(vt::env::REDUCE_ORDER, order);
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

int vt_flat_set_batch_nominate(vt_flat *h, int mode) {
  return guarded([&]() -> int {
  if (!h || (mode != VT_NOMINATE_F32 && mode != VT_NOMINATE_BF16)) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  for (auto &s : h->shards) s->nominate = mode;
  return VT_OK;
  });
}
int vt_flat_batch_nominate(const vt_flat *h) { return h && !h->shards.empty() ? h->shards[0]->nominate : -1; }

int vt_flat_set_batch_shadow(vt_flat *h, int mode) {
  return guarded([&]() -> int {
  if (!h || (mode != VT_SHADOW_OFF && mode != VT_SHADOW_AUTO)) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  for (auto &s : h->shards) {
    s->shadow_mode = mode;
    if (mode == VT_SHADOW_OFF) {  // the room goes back at once
      (void)hipSetDevice(s->ctx.device);
      (void)hipStreamSynchronize(s->ctx.stream);
      s->dShadow.release();
      s->sh_valid = false;
      s->sh_dirty.clear();
    } else {
      s->sh_refused = false;  // asked for again: the next batch looks at the free memory anew
    }
  }
  return VT_OK;
  });
}
int vt_flat_set_single_nominate(vt_flat *h, int enabled) {
  return guarded([&]() -> int {
  if (!h) return VT_ERR_ARGUMENT;
  std::unique_lock<std::shared_mutex> wl(h->rw);
  for (auto &s : h->shards) s->single_nominate = enabled ? 1 : 0;
  return VT_OK;
  });
}
int vt_flat_single_nominate(const vt_flat *h) { return h && !h->shards.empty() ? h->shards[0]->single_nominate.load() : -1; }
int vt_flat_batch_shadow(const vt_flat *h) {
  if (!h || h->shards.empty()) return -1;
  std::shared_lock<std::shared_mutex> rl(h->rw);
  const Shard *ix = h->shards[0].get();
  if (ix->shadow_mode == VT_SHADOW_OFF) return VT_SHADOW_STATE_OFF;
  if (ix->sh_refused) return VT_SHADOW_STATE_REFUSED;
  if (!ix->dShadow.p || !ix->sh_valid) return VT_SHADOW_STATE_NONE;
  return shadow_current(ix) ? VT_SHADOW_STATE_CURRENT : VT_SHADOW_STATE_STALE;
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
  // A bulk load on a one-shard handle checks finiteness beside the copy to the device (index_store_bulk_host:
  // the index is not touched before the whole batch has passed, flat.rs:69-85); everything else here, up front.
  const bool check_beside_copy = !h->multi() && count >= 65536;
  if (check_beside_copy) {
    if (d == 0) return VT_ERR_EMPTY;
    if ((long)d != expected) return VT_ERR_DIMENSION;
  } else {
    VT_TRY(validate_matrix(rows, count, d, expected));
  }
  if (count == 0) return VT_OK;
  RowSource src;
  src.host = rows;
  src.d = d;
  src.unvalidated = check_beside_copy;
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
    st = no_throw([&]() -> int { return shard_delete(ix, id, id_len, &began); });
  } else {
    const size_t s = shard_of(id ? id : "", id_len, h->shards.size());
    st = on_shards(h, std::vector<size_t>{s}, [&](size_t t) -> int {
      return no_throw([&]() -> int { return shard_delete(h->shards[t].get(), id, id_len, &began); });
    });
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
  return coalesced_quantized(h, query, n, candidates, limit, out);
  });
}

int vt_flat_quantized_search_batch(vt_flat *h, const float *queries, size_t nq, size_t d, size_t candidates, size_t limit,
                                   vt_hits **out) {
  return guarded([&]() -> int {
  if (!h || !out || (nq && d && !queries)) return VT_ERR_ARGUMENT;
  return quantized_batch_direct(h, queries, nq, d, candidates, limit, out);
  });
}

int vt_flat_funnel_search(vt_flat *h, const float *query, size_t n, const size_t *stages, size_t nstages,
                          size_t candidates, size_t limit, vt_hits **out) {
  return guarded([&]() -> int {
  if (!h || !out || (!query && n) || (nstages && !stages)) return VT_ERR_ARGUMENT;
  *out = nullptr;
  return coalesced_funnel(h, query, n, stages, nstages, candidates, limit, out);
  });
}

int vt_flat_funnel_search_batch(vt_flat *h, const float *queries, size_t nq, size_t d, const size_t *stages, size_t nstages,
                                size_t candidates, size_t limit, vt_hits **out) {
  return guarded([&]() -> int {
  if (!h || !out || (nq && d && !queries) || (nstages && !stages)) return VT_ERR_ARGUMENT;
  return funnel_batch_direct(h, queries, nq, d, stages, nstages, candidates, limit, out);
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
      a.q = c.qsrc;
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
      j.order = default_order();
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
  std::vector<uint64_t> qwords(2 * (size_t)pairs, 0);  // (K4 reads whole word pairs: an odd count is padded with a zero word)
  std::copy(query, query + W, qwords.begin());
  VT_TRY(dQ.ensure(qwords.size()));
  VT_TRY(dRank.ensure(n));
  VT_HIP(hipMemcpyAsync(dBits.p, packed.data(), packed.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c.stream));
  VT_HIP(hipMemcpyAsync(dQ.p, qwords.data(), qwords.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c.stream));
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
      t.nominate_launches += p.nominate_launches;
      t.nominate_ms += p.nominate_ms;
      t.nominate_bytes += p.nominate_bytes;
      t.nominate_flops += p.nominate_flops;
      t.nominate_queries += p.nominate_queries;
      t.nominate_second_passes += p.nominate_second_passes;
      t.nominate_candidates += p.nominate_candidates;
      t.hamming_queries += p.hamming_queries;
      t.hybrid_device_chains += p.hybrid_device_chains;
      t.prefix_queries += p.prefix_queries;
      t.nominate_shadow_launches += p.nominate_shadow_launches;
      t.shadow_builds += p.shadow_builds;
      t.shadow_build_ms += p.shadow_build_ms;
      t.shadow_patched_rows += p.shadow_patched_rows;
      t.sweep_queries += p.sweep_queries;
      if (reset) c.prof = vt_profile{};
    });
  if (reset) h->xprof = vt_profile{};
  *out = t;
  return VT_OK;
  });
}

int vt_flat_get_profile_sized(vt_flat *h, void *out, size_t out_bytes, int reset) {
  if (!out) return VT_ERR_ARGUMENT;
  vt_profile t{};
  const int st = vt_flat_get_profile(h, &t, reset);
  if (st == VT_OK) std::memcpy(out, &t, std::min(out_bytes, sizeof t));
  return st;
}

}  // extern "C"
