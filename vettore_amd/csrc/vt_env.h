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

// X(key, "name", parse, default)            environment name = "VT_" + upper-case name
#define VT_ENV_PRODUCT_KEYS(X)                                                                                          \
  /* configuration of new indexes */                                                                                    \
  X(REDUCE_ORDER, "reduce_order", P_ORDER, 3)                                                                             \
  X(BATCH_NOMINATE, "batch_nominate", P_NOMINATE, 2)                                                                      \
  X(BATCH_SHADOW, "batch_shadow", P_SHADOW, 1)                                                                            \
  X(SINGLE_NOMINATE, "single_nominate", P_FLAG, 0)                                                                        \
  X(BF16_MIN_RANK, "bf16_min_rank", P_INT, 6)                                                                             \
  X(SLAB, "slab", P_SLAB, 0)                                                                                              \
  X(SLAB_CHUNK_MB, "slab_chunk_mb", P_INT, 0)                                                                             \
  X(BLOCKS_PER_CU, "blocks_per_cu", P_INT, 0)                                                                             \
  X(HAMMING_BLOCKS_PER_CU, "hamming_blocks_per_cu", P_INT, 0)                                                             \
  /* many callers on one handle */                                                                                      \
  X(COALESCE, "coalesce", P_INT, 1)                                                                                       \
  X(COALESCE_SLOTS, "coalesce_slots", P_INT, 0)                                                                           \
  X(COALESCE_GATHER, "coalesce_gather", P_INT, 1)                                                                         \
  /* multi-shard handles */                                                                                             \
  X(SHARD_EXCHANGE, "shard_exchange", P_EXCHANGE, 0)                                                                      \
  X(SHARD_FORCE_WORKERS, "shard_force_workers", P_FLAG, 0)                                                                \
  X(EXCHANGE_TIMEOUT_MS, "exchange_timeout_ms", P_INT, 20000)                                                             \
  X(STAGED_ROUNDS, "staged_rounds", P_FLAG, 0)                                                                            \
  X(LOG, "log", P_FLAG, 0)                                                                                                \
  /* A/B switches of the search paths (DESIGN_APPENDIX A.10) */                                                         \
  X(BATCH_NO_MFMA, "batch_no_mfma", P_FLAG, 0)                                                                            \
  X(NO_MULTI_SCAN, "no_multi_scan", P_FLAG, 0)                                                                            \
  X(NO_SWEEP_GROUPS, "no_sweep_groups", P_FLAG, 0)                                                                        \
  X(NO_GROUP_PIPELINE, "no_group_pipeline", P_FLAG, 0)                                                                    \
  X(QGROUP_STREAMS, "qgroup_streams", P_INT, 0)                                                                           \
  X(NO_PATTERN_BITS, "no_pattern_bits", P_FLAG, 0)                                                                        \
  X(NO_PATTERN_GROUPS, "no_pattern_groups", P_FLAG, 0)                                                                    \
  X(NO_QUANTIZED_GROUPS, "no_quantized_groups", P_FLAG, 0)                                                                \
  X(NO_FUNNEL_GROUPS, "no_funnel_groups", P_FLAG, 0)                                                                      \
  X(FUNNEL_DENSE_SAMPLE, "funnel_dense_sample", P_FLAG, 0)                                                                \
  X(NO_THRESHOLD_SELECT, "no_threshold_select", P_FLAG, 0)                                                                \
  X(HAMMING_LISTS, "hamming_lists", P_FLAG, 0)                                                                            \
  X(HYBRID_CHAIN, "hybrid_chain", P_FLAG, 0)                                                                              \
  X(EAGER_RANKS, "eager_ranks", P_FLAG, 0)                                                                                \
  X(RESCORE_BLOCKS, "rescore_blocks", P_INT, 8)                                                                           \
  X(BATCH_TAIL_CUS, "batch_tail_cus", P_INT, -1)                                                                          \
  X(BATCH_SAMPLE_TILES, "batch_sample_tiles", P_INT, 0)                                                                   \
  X(BATCH_PASS_FIVE, "batch_pass_five", P_FLAG, 0)                                                                        \
  X(DIRECT_QUERY, "direct_query", P_FLAG, 0)                                                                              \
  X(PM_PANEL, "pm_panel", P_INT, 0)                                                                                       \
  X(PM_BLOCKS, "pm_blocks", P_INT, 0)                                                                                     \
  X(CS_PANEL, "cs_panel", P_INT, 0)                                                                                       \
  X(SHADOW_STAGES, "shadow_stages", P_INT, 5)                                                                             \
  X(BATCH_KERNEL, "batch_kernel", P_INT, 0)                                                                               \
  X(BATCH_DEBUG, "batch_debug", P_INT, 0)                                                                                 \
  X(MQ_DBG, "mq_dbg", P_INT, 0)                                                                                           \
  X(SCAN_RT_ORDER, "scan_rt_order", P_FLAG, 0)                                                                            \
  X(MULTI_NO_SLIM, "multi_no_slim", P_FLAG, 0)                                                                            \
  X(MULTI_GENERAL, "multi_general", P_FLAG, 0)                                                                            \
  X(MULTI_NO_PACK, "multi_no_pack", P_FLAG, 0)                                                                            \
  /* ingest (A/B) */                                                                                                    \
  X(INGEST_SERIAL, "ingest_serial", P_FLAG, 0)                                                                            \
  X(INGEST_SEPARATE_CHECK, "ingest_separate_check", P_FLAG, 0)                                                            \
  X(INGEST_STREAMS, "ingest_streams", P_INT, 0)                                                                           \
  X(INGEST_STAGE_MB, "ingest_stage_mb", P_INT, 0)                                                                         \
  /* phases on stderr */                                                                                                \
  X(TRACE_BATCH, "trace_batch", P_FLAG, 0)                                                                                \
  X(TRACE_QGROUP, "trace_qgroup", P_FLAG, 0)                                                                              \
  X(TRACE_HYBRID, "trace_hybrid", P_FLAG, 0)                                                                              \
  X(TRACE_INGEST, "trace_ingest", P_FLAG, 0)                                                                              \
  /* tests only: vt_debug_set, no environment name */                                                                   \
  X(FORCE_BATCH_MFMA, "force_batch_mfma", P_NONE, 0)                                                                      \
  X(FORCE_SWEEP_GROUPS, "force_sweep_groups", P_NONE, 0)                                                                  \
  X(FORCE_MULTI_SCAN, "force_multi_scan", P_NONE, 0)                                                                      \
  X(FORCE_THRESHOLD_SELECT, "force_threshold_select", P_NONE, 0)                                                          \
  X(BF16_RANK, "bf16_rank", P_NONE, 0)

// the fault hooks: always in the enumeration (one table layout for every object file of the library), but without an
// environment name and unknown to find() unless the implementation is compiled with -DVT_TEST_HOOKS
#define VT_ENV_HOOK_KEYS(X)                                                                                             \
  X(TEST_EXCHANGE_STALL_MS, "test_exchange_stall_ms", P_INT, 0)                                                           \
  X(TEST_REFUSE_NZBITS, "test_refuse_nzbits", P_FLAG, 0)                                                                  \
  X(TEST_REFUSE_SHADOW, "test_refuse_shadow", P_FLAG, 0)                                                                  \
  X(TEST_FAIL_AFTER_ID_UPDATE, "test_fail_after_id_update", P_FLAG, 0)                                                    \
  X(TEST_INGEST_LOCKSTEP, "test_ingest_lockstep", P_FLAG, 0)                                                              \
  X(TEST_FOREIGN_ROWS, "test_foreign_rows", P_FLAG, 0)                                                                    \
  X(TEST_COALESCE_HOLD_UNTIL, "test_coalesce_hold_until", P_INT, 0)

enum Key : int {
#define VT_ENV_ENUM(key, name, parse, dflt) key,
  VT_ENV_PRODUCT_KEYS(VT_ENV_ENUM)
  kProductCount,
  kBeforeHooks = kProductCount - 1,  // (the first hook key takes the value kProductCount: the table has no hole)
  VT_ENV_HOOK_KEYS(VT_ENV_ENUM)
#undef VT_ENV_ENUM
  kCount
};

// One definition per loaded library (vt_index.cpp; a stand-alone program that includes a .hip file of the library, or
// tests/concurrency_check.cpp, defines VT_ENV_IMPLEMENTATION before including this header).
long get(Key k);              // an atomic load
void set(Key k, long value);  // an atomic store
int find(const char *name);   // by name (vt_debug_set / vt_debug_get): -1 = no such setting in this build
const char *name_of(int k);
inline bool on(Key k) { return get(k) != 0; }

#ifdef VT_ENV_IMPLEMENTATION

struct Spec {
  const char *name;
  Parse parse;
  long dflt;
};

static const Spec kSpecs[] = {
#define VT_ENV_SPEC(key, name, parse, dflt) {name, parse, dflt},
    VT_ENV_PRODUCT_KEYS(VT_ENV_SPEC)
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
    for (int k = 0; k < kCount; ++k) v[k].store(0, std::memory_order_relaxed);
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
      v[k].store(parse_value(s.parse, e, s.dflt), std::memory_order_relaxed);
    }
  }
};

static Table &table() {
  static Table t;
  return t;
}
// (filled when the library's initialisers run, i.e. inside dlopen / :erlang.load_nif -- not by whichever search comes first)
static const int kTableLoaded = (table(), 0);

long get(Key k) { return table().v[k].load(std::memory_order_relaxed); }
void set(Key k, long value) { table().v[k].store(value, std::memory_order_relaxed); }
int find(const char *name) {
  if (!name) return -1;
  for (int k = 0; k < kKnown; ++k)
    if (!std::strcmp(kSpecs[k].name, name)) return k;
  return -1;
}
const char *name_of(int k) {
  return k >= 0 && k < kKnown ? kSpecs[k].name : nullptr;
}

#endif  // VT_ENV_IMPLEMENTATION

}  // namespace env
}  // namespace vt
