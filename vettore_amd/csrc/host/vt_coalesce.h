// vt_coalesce.h -- searches that meet on one handle travel as one batch.
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// ---- searches that meet on one handle go together -----------------------------------------
// The reference's readers share an RwLock and scale with the host's cores.  Here every search
// is a pass over the corpus in HBM, and callers that run side by side on their own streams
// each read all of it.  So a search that finds another one running waits for it; whoever waits
// first then leads everything that has queued up with its limit as ONE batch -- the batch path
// gives every query the hits its own search would get, bit for bit -- and the others wake up
// with their lists.  An idle handle adds nothing: the first caller runs at once, alone.  Small
// corpora (latency-bound, not bandwidth-bound) keep more than one operation in flight
// (coalesce_slots).  What would force work a lone search avoids (a strict re-rank after unsorted inserts) is not batched: those
// callers are released to search side by side as before.  `VT_COALESCE=0` switches it off.
// The queueing itself is vt_host::coalesced_search_t (host/vt_concurrency.h, checked under
// ThreadSanitizer with stub operations by tests/test_concurrency.py); here are the operations.
// Operations in flight before callers start to queue: a pass over a large corpus owns the
// memory system, two over a medium one still overlap their fixed costs, and searches of a
// corpus of a few MB are all fixed cost -- there every reader context runs side by side and
// only the callers beyond them travel together (tools/reader_probe.cpp).
unsigned coalesce_slots(uint64_t corpus_bytes) {
  if (const long v = vt::env::get(vt::env::COALESCE_SLOTS)) return (unsigned)std::max<long>(1, v);
  return corpus_bytes < (64ull << 20) ? (unsigned)kMaxContexts : corpus_bytes < (1ull << 30) ? 2u : 1u;
}

bool coalescing_enabled() {
  return vt::env::on(vt::env::COALESCE);  // (VT_COALESCE=0)
}

// vt_host::Waiting::kind; aux = candidates (quantized) / index into the handle's funnel shapes (funnel)
enum { COALESCE_SEARCH = 0, COALESCE_QUANTIZED = 1, COALESCE_FUNNEL = 2 };

vt_flat::FunnelShape funnel_shape(vt_flat *h, size_t index) {
  std::lock_guard<std::mutex> g(h->funnel_mu);
  return h->funnel_shapes[index];
}

struct CoalesceOps {
  static constexpr int kOutOfMemory = VT_ERR_NOMEM;
  static vt_host::Coalescer &coalescer(vt_flat *h) { return h->co; }
  static unsigned slots(vt_flat *h) { return coalesce_slots(h->approx_bytes.load(std::memory_order_relaxed)); }
  static int search_direct(vt_flat *h, int kind, size_t aux, const float *query, size_t n, size_t limit, vt_hits **out) {
    if (kind == COALESCE_QUANTIZED) return quantized_direct(h, query, n, aux, limit, out);
    if (kind == COALESCE_FUNNEL) {
      const vt_flat::FunnelShape f = funnel_shape(h, aux);
      return funnel_direct(h, query, n, f.stages.data(), f.stages.size(), f.candidates, limit, out);
    }
    return ::search_direct(h, query, n, limit, out);
  }
  static void search_alone(vt_flat *h, vt_host::Waiting *w) {
    w->status = search_direct(h, w->kind, w->aux, w->query, w->n, w->limit, w->out);
    if (w->status != VT_OK) w->error = g_last_error;
  }
  // every query is judged on its own (flat.rs:97-101), as if it had come alone
  static void judge(vt_flat *h, std::vector<vt_host::Waiting *> &members, std::vector<vt_host::Waiting *> *good) {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    const long dim = handle_dim(h);
    for (vt_host::Waiting *w : members) {
      const int st = h->poisoned ? poisoned_status() : validate_vector(w->query, w->n, dim);
      if (st != VT_OK) {
        w->status = st;
        w->error = st == VT_ERR_POISONED ? g_last_error : std::string();
      } else {
        good->push_back(w);
      }
    }
  }
  static int batch(vt_flat *h, int kind, size_t aux, const float *queries, size_t nq, size_t n, size_t limit, vt_hits **outs) {
    if (kind == COALESCE_QUANTIZED) return quantized_batch_direct(h, queries, nq, n, aux, limit, outs);
    if (kind == COALESCE_FUNNEL) {
      const vt_flat::FunnelShape f = funnel_shape(h, aux);
      return funnel_batch_direct(h, queries, nq, n, f.stages.data(), f.stages.size(), f.candidates, limit, outs);
    }
    return batch_direct(h, queries, nq, n, limit, outs);
  }
  // a batch needs strictly current id ranks; a lone search after unsorted inserts does not
  // (a quantized search needs them either way: nothing to spare by disbanding)
  static bool must_disband(vt_flat *h, int kind, size_t limit) {
    if (h->multi() || kind != COALESCE_SEARCH) return false;
    std::shared_lock<std::shared_mutex> rl(h->rw);
    return shard_stale(h->shards[0].get(), NEED_STRICT_RANKS, limit);
  }
  // a matrix-core pass carries up to 256 plain searches of the dot / L2 family; everything else
  // rides in groups of eight per sweep (K1m, the grouped Hamming and prefix passes)
  static size_t capacity(vt_flat *h, int kind) {
    if (kind != COALESCE_SEARCH) return 8;
    const int m = h->metric;
    const bool gemm = m == VT_COSINE || m == VT_INNER_PRODUCT || m == VT_NEG_INNER_PRODUCT || m == VT_L2 || m == VT_L2_SQUARED;
    return gemm && !vt::env::on(vt::env::BATCH_NO_MFMA) ? 256 : 8;
  }
  // tests that need callers to MEET (tests/test_gpu_coalesce.py, libvettore_hip_hooks.so only): the product never holds
#ifdef VT_TEST_HOOKS
  static size_t hold_until(vt_flat *) { return (size_t)std::max<long>(0, vt::env::get(vt::env::TEST_COALESCE_HOLD_UNTIL)); }
#else
  static constexpr size_t hold_until(vt_flat *) { return 0; }
#endif
  static void run(vt_flat *h, std::vector<vt_host::Waiting *> &members) { vt_host::run_coalesced_t<vt_flat, CoalesceOps>(h, members); }
  static void drop_hits(vt_hits *hits) { delete hits; }
  static void set_last_error(const std::string &msg) { g_last_error = msg; }
};

int coalesced_search(vt_flat *h, const float *query, size_t n, size_t limit, vt_hits **out) {
  if (limit == 0 || limit > (size_t)vt::kMaxFusedK || n == 0 || !coalescing_enabled()) return search_direct(h, query, n, limit, out);
  // float hamming / jaccard on a corpus with its non-zero-bit column: callers whose lists fit K4p's
  // wave buffers share sweeps of the column like everyone else; longer lists are one short K4 pass
  // (or a few) per caller, side by side on the reader contexts
  if (limit > (size_t)vt::kSmallK && pattern_metric(h->shards[0]->metric) &&
      h->approx_rows.load(std::memory_order_relaxed) >= kPatternMinRows)
    return search_direct(h, query, n, limit, out);
  return vt_host::coalesced_search_t<vt_flat, CoalesceOps>(h, query, n, limit, out);
}

// quantized_search callers that meet on a handle share sweeps of the bit matrix (quantized_group)
// (corpora too small for the grouped passes -- fewer than 16 384 rows -- would be served one by one
// by the batch's leader: their callers stay side by side on the reader contexts instead)
bool too_small_for_groups(vt_flat *h) {
  return h->approx_rows.load(std::memory_order_relaxed) < 16384;
}

int coalesced_quantized(vt_flat *h, const float *query, size_t n, size_t candidates, size_t limit, vt_hits **out) {
  if (limit == 0 || candidates == 0 || candidates > (size_t)vt::kMaxFusedK || n == 0 || h->multi() || !coalescing_enabled() ||
      too_small_for_groups(h))
    return quantized_direct(h, query, n, candidates, limit, out);
  return vt_host::coalesced_search_t<vt_flat, CoalesceOps>(h, query, n, limit, out, COALESCE_QUANTIZED, candidates);
}

// funnel_search callers that meet on a cosine / dot / L2 / L1 / Linf handle share the stage-1 sweep of the prefixes
// (funnel_group): only callers with the same stages and candidates travel together
int coalesced_funnel(vt_flat *h, const float *query, size_t n, const size_t *stages, size_t nstages, size_t candidates,
                     size_t limit, vt_hits **out) {
  bool plain = limit == 0 || candidates == 0 || candidates > (size_t)vt::kMaxFusedK || n == 0 || nstages == 0 || nstages > 16 ||
               h->multi() || !coalescing_enabled() || too_small_for_groups(h) ||
               !(h->shards[0]->metric == VT_COSINE || vt::prefix_multi_supports(h->shards[0]->metric));
  for (size_t i = 0; i < nstages && !plain; ++i) plain = stages[i] == 0 || stages[i] > n;  // (its own error, in its own order)
  size_t shape = 0;
  if (!plain) {
    std::lock_guard<std::mutex> g(h->funnel_mu);
    for (; shape < h->funnel_shapes.size(); ++shape) {
      const vt_flat::FunnelShape &f = h->funnel_shapes[shape];
      if (f.candidates == candidates && f.stages.size() == nstages && std::equal(stages, stages + nstages, f.stages.begin())) break;
    }
    if (shape == h->funnel_shapes.size()) {
      if (shape >= 64) plain = true;  // (a handle asked for more shapes than anyone coalesces over)
      else h->funnel_shapes.push_back(vt_flat::FunnelShape{std::vector<size_t>(stages, stages + nstages), candidates});
    }
  }
  if (plain) return funnel_direct(h, query, n, stages, nstages, candidates, limit, out);
  return vt_host::coalesced_search_t<vt_flat, CoalesceOps>(h, query, n, limit, out, COALESCE_FUNNEL, shape);
}

}  // namespace
