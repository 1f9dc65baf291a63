// vt_env.h -- every switch the library knows, read from the process environment ONCE, when the library is loaded.
//
// Why: the library lives inside a BEAM (nifs.rs:297-309 -- the reference NIF reads no process state at all).  There
// `System.put_env/2` calls setenv on a scheduler thread while dirty-scheduler threads run searches; a search that calls
// getenv to choose its path races with that in glibc.  So: one table of settings, each an atomic integer, filled from
// `VT_<NAME>` by the one getenv loop below when the shared object's initialisers run (vt_index.cpp forces that), and
// never again.  Afterwards a setting changes only through vt_debug_set (include/vettore_flat.h) -- an atomic store.
//
// Switches whose only users are tests (force_* : take a path on corpora the cost model would never send there) have NO
// environment name: the product library does not contain the strings, tests reach them through vt_debug_set.  The fault
// hooks (test_*) exist only in builds with -DVT_TEST_HOOKS (libvettore_hip_hooks.so).
//
// Stand-alone on purpose (no HIP, no other header of the library): the device launchers include it, the probes under
// tools/ that include a .hip file compile their own copy (VT_ENV_IMPLEMENTATION), and tests/concurrency_check.cpp runs
// get() beside a thread that calls setenv under ThreadSanitizer.
#pragma once

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace vt {
namespace env {

// how a value is read off the environment string
enum Parse {
  P_FLAG,      // unset, "" or "0" -> 0; a number -> that number; anything else -> 1
  P_INT,       // atol (unset -> the default)
  P_ORDER,     // pair | avx | seq | sse2 (or 0..3)
  P_NOMINATE,  // f32 | 1 -> 1 (VT_NOMINATE_F32), else 2 (VT_NOMINATE_BF16)
  P_SHADOW,    // 0 | off -> 0 (VT_SHADOW_OFF), else 1 (VT_SHADOW_AUTO)
  P_SLAB,      // malloc -> 1, else 0
  P_EXCHANGE,  // host -> 1, rccl -> 2, else 0 (the handle chooses)
  P_NONE,      // no environment name: vt_debug_set only
};

// X(key, "name", parse, default, "what it does")   environment name = "VT_" + upper-case name; the last argument is
// documentation only (tools/env_table.py writes DESIGN_APPENDIX A.10 from it): no expansion below uses it
//
// r06: this list is what a maintainer has to trust -- every entry chooses between paths that are ALIVE in the product
// (configuration, or forcing a path the cost model takes anyway on other shapes, which is what the parity tests use them
// for).  The A/B switches of rounds 2-5 whose alternative lost (DESIGN_APPENDIX A.10 lists them with the measurement that
// retired each) are gone, and so are the losing paths: nifs.rs has zero knobs.
#define VT_ENV_PRODUCT_KEYS(X) \
  /* configuration of new indexes */ \
  X(REDUCE_ORDER, "reduce_order", P_ORDER, 3, "lane order of `wide::f32x8::reduce_add` for indexes created afterwards: `pair`, `avx`, `seq`, `sse2` (default; DESIGN 3.3); per handle: `vt_flat_set_reduce_order`") \
  X(BATCH_NOMINATE, "batch_nominate", P_NOMINATE, 2, "`f32`: batches nominate on the FP32 matrix cores (K2) instead of the bf16 ones (K2b / K2s, the default); per handle: `vt_flat_set_batch_nominate`") \
  X(BATCH_SHADOW, "batch_shadow", P_SHADOW, 1, "`0` / `off`: never build the bf16 shadow of the rows (K2b from the f32 rows instead of K2s); per handle: `vt_flat_set_batch_shadow`") \
  X(SINGLE_NOMINATE, "single_nominate", P_FLAG, 0, "`1`: lone searches go through the shadow like a batch of one (opt-in); per handle: `vt_flat_set_single_nominate`") \
  X(BF16_MIN_RANK, "bf16_min_rank", P_INT, 6, "K2b / K2s: smallest sample rank the threshold is taken from (default 6)") \
  X(SLAB, "slab", P_SLAB, 0, "`malloc`: the row slab as one allocation regrown by copy -- the form slabs below one chunk take anyway (`tools/capacity_probe.py`)") \
  X(SLAB_CHUNK_MB, "slab_chunk_mb", P_INT, 0, "chunk size of the mapped slab (default 1 GiB; tests use 2 MiB so that small corpora cross chunk borders)") \
  X(INGEST_STAGE_MB, "ingest_stage_mb", P_INT, 0, "size of a pinned staging quarter of a bulk load (default 128; tests use 1 so that small loads cross many)") \
  /* many callers on one handle */ \
  X(COALESCE, "coalesce", P_INT, 1, "`0`: searches on one handle never wait for each other (side by side on their own streams)") \
  X(COALESCE_SLOTS, "coalesce_slots", P_INT, 0, "operations in flight on a handle before callers queue (default by corpus size, DESIGN 6.3; tests use 1)") \
  /* multi-shard handles */ \
  X(SHARD_EXCHANGE, "shard_exchange", P_EXCHANGE, 0, "multi-shard handles: `host` / `rccl` -- how the shards' lists meet (default: RCCL when every shard has its own device)") \
  X(SHARD_FORCE_WORKERS, "shard_force_workers", P_FLAG, 0, "a one-shard handle goes through the worker / exchange machinery (tests, `bench.py --gpus 1 --exchange ...`)") \
  X(EXCHANGE_TIMEOUT_MS, "exchange_timeout_ms", P_INT, 20000, "deadline behind a shard's all-gather (default 20 000): a wedged exchange fails the search with a message") \
  X(STAGED_ROUNDS, "staged_rounds", P_FLAG, 0, "multi-shard handles: staged searches as one fan-out per stage (the form a failed one-round attempt falls back to) instead of one per search") \
  X(LOG, "log", P_FLAG, 0, "the handle's exchange note on stderr at creation") \
  X(TRACE_INGEST, "trace_ingest", P_FLAG, 0, "`1`: a bulk store prints its phases on stderr; `2`: every store call does (`tools/insert_probe.py`)") \
  /* never take a path the cost model would (the other one is alive: other shapes take it) */ \
  X(BATCH_NO_MFMA, "batch_no_mfma", P_FLAG, 0, "batches never take the shared matrix-core pass (K1m / K1p sweeps or single scans instead: what the other metrics take)") \
  X(NO_MULTI_SCAN, "no_multi_scan", P_FLAG, 0, "batches never take K1m (K2 or single scans instead)") \
  X(NO_GROUP_PIPELINE, "no_group_pipeline", P_FLAG, 0, "the groups of one batch call are waited for one by one (what a call takes when no second context is to be had, or while profiling)") \
  /* tests only: always take a path, on corpora the cost model would never send there (vt_debug_set, no environment name) */ \
  X(FORCE_BATCH_MFMA, "force_batch_mfma", P_NONE, 0, "tests: the shared matrix-core pass whatever the cost model says (small corpora)") \
  X(FORCE_SWEEP_GROUPS, "force_sweep_groups", P_NONE, 0, "tests and soaks: K1p sweeps on corpora of a few MB") \
  X(FORCE_MULTI_SCAN, "force_multi_scan", P_NONE, 0, "tests: K1m on corpora of a few thousand rows") \
  X(FORCE_THRESHOLD_SELECT, "force_threshold_select", P_NONE, 0, "tests: the key-column threshold path for limits 257..4 096 on small corpora") \
  X(BF16_RANK, "bf16_rank", P_NONE, 0, "tests: K2b / K2s take their threshold from exactly this sample rank (`= limit` leaves no margin: every query takes the second pass)")

// Timing experiments (wrong results on purpose: barriers removed, stages skipped) and phase traces of the batch paths:
// always in the enumeration, but named, read and stored only in builds made with -DVT_EXPERIMENTS (`make experiments`:
// vettore_amd/lib/experiments/libvettore_hip.so).  In the product get() of one of these is its default, a constant the
// compiler folds: the code behind it is not in the library.
#define VT_ENV_EXPERIMENT_KEYS(X) \
  X(BATCH_DEBUG, "batch_debug", P_INT, 0, "K2's timing experiments (`tools/batch_debug.sh`): bits remove barriers / waits / the candidate append") \
  X(MQ_DBG, "mq_dbg", P_INT, 0, "K1m's timing experiments (cost breakdown of DESIGN_APPENDIX A.4)") \
  X(SHADOW_STAGES, "shadow_stages", P_INT, 5, "depth of K2s's LDS ring for lone passes, 4 or 5 (default 5; pipelined groups run on 4)") \
  X(TRACE_BATCH, "trace_batch", P_FLAG, 0, "a batch group prints its phases on stderr")

// the fault hooks: always in the enumeration (one table layout for every object file of the library), but without an
// environment name and unknown to find() unless the implementation is compiled with -DVT_TEST_HOOKS
#define VT_ENV_HOOK_KEYS(X) \
  X(TEST_EXCHANGE_STALL_MS, "test_exchange_stall_ms", P_INT, 0, "a shard's all-gather stalls this long (the timeout path)") \
  X(TEST_REFUSE_NZBITS, "test_refuse_nzbits", P_FLAG, 0, "the non-zero-bit column is refused as if the card were full") \
  X(TEST_REFUSE_SHADOW, "test_refuse_shadow", P_FLAG, 0, "the bf16 shadow is refused as if the card were full") \
  X(TEST_FAIL_AFTER_ID_UPDATE, "test_fail_after_id_update", P_FLAG, 0, "a mutation fails after it changed the id table (the handle must come out poisoned)") \
  X(TEST_INGEST_LOCKSTEP, "test_ingest_lockstep", P_FLAG, 0, "the id thread of a bulk load keeps step with the verified rows") \
  X(TEST_FOREIGN_ROWS, "test_foreign_rows", P_FLAG, 0, "device-resident rows are treated as living on another GPU (they reach a mapped slab through a staging block)") \
  X(TEST_REFUSE_SPARE_SCRATCH, "test_refuse_spare_scratch", P_FLAG, 0, "the second context of a pipelined batch call cannot get its scratch: the call must go on in series on the first") \
  X(TEST_COALESCE_HOLD_UNTIL, "test_coalesce_hold_until", P_INT, 0, "an idle handle's first caller keeps its slot until this many callers have queued (callers MEET: `vt_callers_meet`)")

enum Key : int {
#define VT_ENV_ENUM(key, name, parse, dflt, doc) key,
  VT_ENV_PRODUCT_KEYS(VT_ENV_ENUM)
  kProductCount,
  kBeforeExperiments = kProductCount - 1,  // (the next key takes the value kProductCount: the table has no hole)
  VT_ENV_EXPERIMENT_KEYS(VT_ENV_ENUM)
  kExperimentEnd,
  kBeforeHooks = kExperimentEnd - 1,
  VT_ENV_HOOK_KEYS(VT_ENV_ENUM)
#undef VT_ENV_ENUM
  kCount
};

constexpr long default_of(Key k) {
  switch (k) {
#define VT_ENV_DEFAULT(key, name, parse, dflt, doc) \
  case key: return dflt;
    VT_ENV_PRODUCT_KEYS(VT_ENV_DEFAULT)
    VT_ENV_EXPERIMENT_KEYS(VT_ENV_DEFAULT)
    VT_ENV_HOOK_KEYS(VT_ENV_DEFAULT)
#undef VT_ENV_DEFAULT
    default: return 0;
  }
}

// One definition per loaded library (vt_index.cpp; a stand-alone program that includes a .hip file of the library, or
// tests/concurrency_check.cpp, defines VT_ENV_IMPLEMENTATION before including this header).
long load(Key k);             // an atomic load
inline long get(Key k) {
#ifndef VT_EXPERIMENTS
  if (k >= kProductCount && k < kExperimentEnd) return default_of(k);  // (k is a constant at every call: folded)
#endif
  return load(k);
}
void set(Key k, long value);  // an atomic store
int find(const char *name);   // by name (vt_debug_set / vt_debug_get): the Key, or -1 = no such setting in this build
bool valid(int k, long value);  // what the setting's parser could have produced (vt_debug_set refuses anything else)
const char *name_of(int k);
inline bool on(Key k) { return get(k) != 0; }

#ifdef VT_ENV_IMPLEMENTATION

struct Spec {
  Key key;
  const char *name;
  Parse parse;
  long dflt;
};

static const Spec kSpecs[] = {
#define VT_ENV_SPEC(key, name, parse, dflt, doc) {key, name, parse, dflt},
    VT_ENV_PRODUCT_KEYS(VT_ENV_SPEC)
#ifdef VT_EXPERIMENTS
    VT_ENV_EXPERIMENT_KEYS(VT_ENV_SPEC)
#endif
#ifdef VT_TEST_HOOKS
    VT_ENV_HOOK_KEYS(VT_ENV_SPEC)
#endif
#undef VT_ENV_SPEC
};
constexpr int kKnown = (int)(sizeof kSpecs / sizeof kSpecs[0]);  // settings this build reads and names

static long parse_value(Parse p, const char *e, long dflt) {
  if (!e) return dflt;
  switch (p) {
    case P_FLAG: {
      if (e[0] == '\0' || (e[0] == '0' && e[1] == '\0')) return 0;
      char *end = nullptr;
      const long v = std::strtol(e, &end, 10);
      return end != e && *end == '\0' ? v : 1;
    }
    case P_INT: {
      char *end = nullptr;
      const long v = std::strtol(e, &end, 10);
      return end != e ? v : dflt;
    }
    case P_ORDER:
      if (!std::strcmp(e, "pair") || !std::strcmp(e, "0")) return 0;
      if (!std::strcmp(e, "avx") || !std::strcmp(e, "1")) return 1;
      if (!std::strcmp(e, "seq") || !std::strcmp(e, "2")) return 2;
      if (!std::strcmp(e, "sse2") || !std::strcmp(e, "3")) return 3;
      return dflt;
    case P_NOMINATE: return !std::strcmp(e, "f32") || !std::strcmp(e, "1") ? 1 : 2;
    case P_SHADOW: return !std::strcmp(e, "0") || !std::strcmp(e, "off") ? 0 : 1;
    case P_SLAB: return !std::strcmp(e, "malloc") ? 1 : 0;
    case P_EXCHANGE: return !std::strcmp(e, "host") ? 1 : !std::strcmp(e, "rccl") ? 2 : 0;
    case P_NONE: return dflt;
  }
  return dflt;
}

struct Table {
  std::atomic<long> v[kCount];
  Table() {
    for (int k = 0; k < kCount; ++k) v[k].store(default_of((Key)k), std::memory_order_relaxed);
    // THE read of the environment: once per loaded library, on the loading thread
    for (int k = 0; k < kKnown; ++k) {
      const Spec &s = kSpecs[k];
      const char *e = nullptr;
      if (s.parse != P_NONE) {
        char var[64] = "VT_";
        size_t n = 3;
        for (const char *p = s.name; *p && n + 1 < sizeof var; ++p) var[n++] = (char)(*p >= 'a' && *p <= 'z' ? *p - 'a' + 'A' : *p);
        var[n] = '\0';
        e = std::getenv(var);
      }
      v[s.key].store(parse_value(s.parse, e, s.dflt), std::memory_order_relaxed);
    }
  }
};

static Table &table() {
  static Table t;
  return t;
}
// (filled when the library's initialisers run, i.e. inside dlopen / :erlang.load_nif -- not by whichever search comes first)
static const int kTableLoaded = (table(), 0);

long load(Key k) { return table().v[k].load(std::memory_order_relaxed); }
void set(Key k, long value) { table().v[k].store(value, std::memory_order_relaxed); }
static const Spec *spec_of(int k) {
  for (int i = 0; i < kKnown; ++i)
    if (kSpecs[i].key == k) return &kSpecs[i];
  return nullptr;
}
int find(const char *name) {
  if (!name) return -1;
  for (int i = 0; i < kKnown; ++i)
    if (!std::strcmp(kSpecs[i].name, name)) return kSpecs[i].key;
  return -1;
}
bool valid(int k, long value) {
  const Spec *s = spec_of(k);
  if (!s) return false;
  switch (s->parse) {
    case P_ORDER: return value >= 0 && value <= 3;
    case P_NOMINATE: return value == 1 || value == 2;
    case P_SHADOW:
    case P_SLAB: return value == 0 || value == 1;
    case P_EXCHANGE: return value >= 0 && value <= 2;
    default: return true;
  }
}
const char *name_of(int k) {
  const Spec *s = spec_of(k);
  return s ? s->name : nullptr;
}

#endif  // VT_ENV_IMPLEMENTATION

}  // namespace env
}  // namespace vt
