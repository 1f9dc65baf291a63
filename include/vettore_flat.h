/*
 * vettore_flat.h -- C ABI of libvettore_hip.so, the MI355X (gfx950) drop-in for
 * the flat-index hot path of elchemista/vettore v0.3.2.
 *
 * Every entry point is what a NIF shim for `Vettore.Nifs` (or the plugin module
 * `Vettore.Index.FlatGpu`) binds in place of the Rust function
 * cited next to it.  Citations are paths under /root/reference.
 *
 * Conventions
 *  - plain pointers and sizes only; no exceptions cross the boundary;
 *  - every function returns a VT_* status; vt_strerror() maps the reference's
 *    error statuses to the reference's exact error strings, so the shim can
 *    build `{:error, "dimension mismatch"}` verbatim;
 *  - ids are arbitrary byte strings (Elixir binaries): pointer + length, or for
 *    batches one concatenated buffer plus `off[count + 1]` offsets;
 *  - ragged vector batches use the same layout (values + `off[count + 1]` in
 *    elements) so that a wrong-length row is reported as "dimension mismatch"
 *    exactly like native/vettore/src/flat.rs:69-85 does;
 *  - all compute runs on the GPU; there is no CPU fallback.  Without a usable
 *    HIP device the calls fail with VT_ERR_DEVICE.
 *  - thread safety: a handle is the reference's RwLock<FlatIndex> (nifs.rs:266-309):
 *    searches of one handle run concurrently with each other (each on its own HIP
 *    stream; vt_flat_search calls that meet on a busy handle are answered together
 *    in one pass over the corpus, see vt_flat_coalesce_stats), mutations are
 *    exclusive; different handles are independent.  A
 *    mutation that fails on the device after it began changing the index poisons
 *    the handle (VT_ERR_POISONED from then on), like a panic under the write lock.
 */
#ifndef VETTORE_FLAT_H
#define VETTORE_FLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct of this header grows or an entry point changes (r04: vt_profile grew,
 * vt_flat_set_batch_shadow / vt_flat_batch_shadow / vt_flat_get_profile_sized arrived).  A binding
 * compares it with vt_abi_version() when it loads the library (vettore_amd/_lib.py, the erl_nif
 * shim's load callback): a caller built against an older header would hand vt_flat_get_profile a
 * struct that is too small. */
#define VT_ABI_VERSION 4   /* r05: vt_debug_set / vt_debug_get */

/* Metric codes == Metric::from_code, native/vettore/src/distances.rs:24-38,
 * mirrored by lib/vettore/collection.ex:1306-1315. */
enum {
  VT_L2 = 0,
  VT_L2_SQUARED = 1,
  VT_COSINE = 2,
  VT_INNER_PRODUCT = 3,
  VT_NEG_INNER_PRODUCT = 4,
  VT_MANHATTAN = 5,
  VT_CHEBYSHEV = 6,
  VT_HAMMING = 7,
  VT_JACCARD = 8
};

/* Status codes.  1..7 carry the reference's error strings. */
enum {
  VT_OK = 0,
  VT_ERR_EMPTY = 1,          /* "vector must not be empty"            flat.rs:138 */
  VT_ERR_DIMENSION = 2,      /* "dimension mismatch"                  flat.rs:141, distances.rs:44 */
  VT_ERR_NON_FINITE = 3,     /* "vector contains a non-finite value"  distances.rs:135 */
  VT_ERR_OVERFLOW = 4,       /* "metric overflow"                     distances.rs:67 */
  VT_ERR_UNKNOWN_METRIC = 5, /* "unknown metric"                      distances.rs:36 */
  VT_ERR_PREFIX = 6,         /* "invalid prefix dimensions"           search.rs:47 */
  VT_ERR_DIMS_POSITIVE = 7,  /* "dimensions must be positive"         distances.rs:463 */
  VT_ERR_POISONED = 8,       /* "flat lock poisoned"                  nifs.rs:269: an earlier mutation died half-way */
  /* statuses the reference cannot produce */
  VT_ERR_NOMEM = 16,
  VT_ERR_DEVICE = 17,        /* HIP error / no gfx950 device; see vt_last_error() */
  VT_ERR_UNSUPPORTED = 18,   /* shape outside what the device kernels cover; vt_last_error() */
  VT_ERR_ARGUMENT = 19       /* NULL handle and the like (the shim's badarg) */
};

/* Lane order of wide::f32x8::reduce_add used for every 8-float chunk
 * (distances.rs:197-308).  The reference does not pin it (third-party crate,
 * build-flag dependent); the device kernels reproduce whichever is selected bit
 * for bit.  With l0..l7 the lanes of a chunk:
 *   VT_ORDER_PAIR ((l0+l1)+(l2+l3)) + ((l4+l5)+(l6+l7))   two f32x4 halves, SSE3 haddps / NEON vaddvq
 *   VT_ORDER_AVX  ((l0+l4)+(l2+l6)) + ((l1+l5)+(l3+l7))   one 256-bit register (target-cpu=native, Taskfile.yml:12)
 *   VT_ORDER_SEQ  (((l0+l1)+l2)+l3) + (((l4+l5)+l6)+l7)   array fallback (no SIMD target feature)
 *   VT_ORDER_SSE2 ((l0+l2)+(l1+l3)) + ((l4+l6)+(l5+l7))   two f32x4 halves, movehl + shuffle (baseline
 *                                                          x86-64: what the precompiled release NIF is built for)
 * Default VT_ORDER_SSE2 (override: VT_REDUCE_ORDER=pair|avx|seq|sse2, vt_set_default_reduce_order,
 * vt_flat_set_reduce_order).  See DESIGN.md "summation order" and INTEGRATION.md section 4 for the
 * one-line probe that tells which order a given libvettore build uses. */
enum { VT_ORDER_PAIR = 0, VT_ORDER_AVX = 1, VT_ORDER_SEQ = 2, VT_ORDER_SSE2 = 3 };

const char *vt_strerror(int status);
/* Detail text of the calling thread's last VT_ERR_DEVICE / VT_ERR_UNSUPPORTED. */
const char *vt_last_error(void);
int vt_abi_version(void);
/* Number of visible HIP devices (0 if none); never fails. */
int vt_device_count(void);
/* Diagnostic (bench.py's `measured_read_peak`): GB/s of a bare read-only stream over a scratch buffer
 * of `bytes` (random floats; allocated and freed here) on `device`, best of `reps` passes after one
 * warm-up pass, timed with HIP events.  The stream is the LDS-DMA ring the batch passes read through
 * with nothing consuming it (whole 384-KiB tiles, 1-KiB pieces, 64 KiB in flight per CU): the
 * yardstick beside the 8 TB/s spec figure -- a search kernel that reads this way and also computes
 * cannot stream faster on the same box. */
int vt_device_read_peak(int device, size_t bytes, int reps, double *gbps);

/* ------------------------------------------------------------------ settings
 * The library reads the process environment exactly ONCE, while it is being loaded (dlopen /
 * :erlang.load_nif): every VT_* variable of DESIGN_APPENDIX.md A.10 lands in a table of atomic integers
 * (csrc/vt_env.h) and no search, insert or load ever calls getenv -- inside a BEAM `System.put_env/2`
 * may run setenv on a scheduler thread at any time, and the reference NIF itself reads no process state
 * (nifs.rs:297-309).  After loading, a setting changes only through vt_debug_set: `name` is the
 * variable's name without "VT_", lower case ("batch_no_mfma", "coalesce_slots", "slab_chunk_mb" ...;
 * string-valued ones take their code: reduce_order 0..3 = VT_ORDER_*, batch_nominate VT_NOMINATE_*,
 * batch_shadow VT_SHADOW_*, slab 1 = malloc, shard_exchange 1 = host / 2 = rccl).  Four switches have
 * no environment name at all -- "force_batch_mfma", "force_sweep_groups", "force_multi_scan",
 * "force_threshold_select" (take a path on corpora the cost model would never send there) and
 * "bf16_rank" (K2b's threshold from exactly that sample rank): tests and soaks only.  A debugging and
 * testing aid, process-wide, effective for calls that start afterwards; VT_ERR_ARGUMENT for a name this
 * build does not know (the fault hooks "test_*" exist in libvettore_hip_hooks.so only, the timing experiments'
 * switches in `make experiments` only; r06 retired the A/B switches whose alternative had lost -- 24 settings are
 * left, DESIGN_APPENDIX.md A.10) and for a value the setting's own parser could not have produced
 * (reduce_order outside 0..3, batch_nominate outside 1..2 ...). */
int vt_debug_set(const char *name, long value);
int vt_debug_get(const char *name, long *value);

/* ------------------------------------------------------------------ hits
 * Vec<(String, f32)> as returned by flat_search / vector_top_k /
 * binary_top_k: ascending by (rank, id bytes). */
typedef struct vt_hits vt_hits;
size_t vt_hits_len(const vt_hits *hits);
/* id bytes of hit i (owned by `hits`, valid until vt_hits_free). */
const char *vt_hits_id(const vt_hits *hits, size_t i, size_t *len);
/* raw metric value of hit i (the f32 the NIF widens to a BEAM double). */
float vt_hits_raw(const vt_hits *hits, size_t i);
/* f32::total_cmp sort key of rank_value(metric, raw) as an order-preserving
 * u32 (distances.rs:113-119, flat.rs:34-40): what a multi-shard merge compares
 * before falling back to the id bytes. */
uint32_t vt_hits_rank_key(const vt_hits *hits, size_t i);
/* Serialises up to `cap` hits as fixed 64-byte records -- u32 rank_key, f32 raw,
 * u32 id_len, then the first 52 id bytes (zero padded) -- the wire format of the
 * cross-shard merge.  Returns the number of records written. */
#define VT_HIT_RECORD_BYTES 64
#define VT_HIT_RECORD_ID_BYTES 52
size_t vt_hits_pack(const vt_hits *hits, void *records, size_t cap);
/* The same for the `nq` hit lists of a query batch, as the process-per-GPU exchange ships them
 * (vettore_amd/sharded.py): `blocks` receives nq blocks of (limit + 1) records -- record 0 of a block
 * is its header {u32 count, u32 long_ids}: the hits that follow, and whether an id of the block is
 * longer than VT_HIT_RECORD_ID_BYTES (only its first bytes travel inline then). */
void vt_hits_pack_many(const vt_hits *const *hits, size_t nq, size_t limit, void *blocks);
/* ... and the merge on the receiving side: `blocks` holds world x nq such blocks (rank-major, as an
 * all-gather leaves them); out_blocks receives nq blocks with the `limit` best hits of each query over
 * all ranks by (rank key, id bytes) == FlatHit::cmp (flat.rs:34-40) -- one index over all rows would
 * return exactly these.  A block whose header says long_ids is merged by the inline bytes only and keeps
 * the flag: the caller must then order ties by the whole ids itself. */
int vt_hit_blocks_merge(const void *blocks, size_t world, size_t nq, size_t limit, void *out_blocks);
/* All hits in one call (bindings that would otherwise cross the ABI three times per hit):
 * vt_hits_id_bytes = total id bytes; vt_hits_export fills ids (concatenated), id_off
 * (len + 1 offsets), raw and rank_key (each len entries; null pointers are skipped). */
size_t vt_hits_id_bytes(const vt_hits *hits);
void vt_hits_export(const vt_hits *hits, char *ids, size_t *id_off, float *raw, uint32_t *rank_key);
void vt_hits_free(vt_hits *hits);
/* The same for the `n` lists a batch call returned (null entries are skipped): one call across the boundary instead of
 * `n` -- through a foreign-function interface 256 separate frees cost more than the lists took to build (r04). */
void vt_hits_free_many(vt_hits **hits, size_t n);

/* ----------------------------------------------------------- flat index
 * FlatResource(RwLock<FlatIndex>), flat.rs:13-17, :131-134. */
typedef struct vt_flat vt_flat;

/* flat_new_<metric>/0, nifs.rs:200-257.  `device` = HIP device ordinal. */
int vt_flat_new(int metric_code, int device, vt_flat **out);
/* The same resource spread over the GPUs of one node (SURVEY.md 8b "device_mask", 8e):
 * ONE handle in ONE process, like the reference's FlatResource (nifs.rs:254-257); shard i
 * lives on HIP device devices[i].  A row belongs to shard hash(id) % ndev, so every
 * operation on an id goes to one shard and the shards never talk to each other except
 * for the per-query exchange of their top-k lists.  Every entry point below accepts the
 * handle; results equal those of a one-GPU index over all rows (ids, order, raw bits).
 * ndev == 1 is vt_flat_new.  The same ordinal may be listed more than once (tests on a
 * one-GPU box); the exchange then stays on the host path. */
int vt_flat_new_sharded(int metric_code, const int *devices, size_t ndev, vt_flat **out);
size_t vt_flat_shard_count(const vt_flat *index);
int vt_flat_shard_device(const vt_flat *index, size_t shard);  /* -1: no such shard */
size_t vt_flat_shard_len(const vt_flat *index, size_t shard);
/* Where a shard's rows live: rows the slab holds without growing, its bytes, and the number of
 * physical chunks mapped into its reserved range (0: a plain allocation, the form below one
 * chunk).  A slab of one chunk or more grows by mapping further 1-GiB chunks behind the rows --
 * they never move and no second slab exists beside the first (the reference's per-row Vec
 * allocations, flat.rs:13-17, have no such step at all).  Any out pointer may be NULL. */
int vt_flat_shard_memory(const vt_flat *index, size_t shard, size_t *row_capacity, size_t *slab_bytes,
                         size_t *slab_chunks);
/* Searches that meet on one handle go together.  The reference's readers share an RwLock and
 * scale with the host's cores (nifs.rs:297-309); here every search is a pass over the corpus in
 * HBM, so a vt_flat_search that finds another one running waits for it, and everything that has
 * queued up by then (same limit, up to 256) runs as ONE batch -- one sweep of the corpus for up
 * to eight queries, the matrix-core pass beyond -- each caller getting exactly the hits its own
 * search would have produced (ids, order, raw bits).  An idle handle adds nothing: the first
 * caller runs at once, alone.  VT_COALESCE=0 in the environment switches it off.  This returns
 * how many such batches ran and how many searches they carried.  Out pointers may be NULL. */
int vt_flat_coalesce_stats(vt_flat *index, uint64_t *batches, uint64_t *batched_queries);
/* out_shard[i] = shard that owns (or would own) id i. */
int vt_flat_route_ids(const vt_flat *index, size_t count, const char *ids, const size_t *id_off,
                      uint32_t *out_shard);
/* How the shards' top-k lists meet (multi-shard handles):
 *   VT_EXCHANGE_RCCL  every shard's select kernel leaves its list in a device block; one
 *                     ncclAllGather per shard communicator (RCCL over xGMI, queued behind the
 *                     scan on the shard's stream) collects the blocks, shard 0 hands the
 *                     gathered lists to the host, which merges by (rank key, id bytes).
 *                     Default when the shards sit on distinct devices and limit <= 256.
 *   VT_EXCHANGE_HOST  every shard's select kernel writes its list straight into pinned host
 *                     memory; same merge.  Always available; used for limits above 256,
 *                     batches, and when RCCL cannot serve the device list. */
enum { VT_EXCHANGE_HOST = 0, VT_EXCHANGE_RCCL = 1 };
int vt_flat_set_exchange(vt_flat *index, int mode);
int vt_flat_exchange(const vt_flat *index);
/* Which exchange a multi-shard handle chose when it was created and, if RCCL was refused, why
 * (one line, also printed on stderr at creation when VT_LOG is set).  Empty for a plain index.
 * A search whose all-gather does not complete within VT_EXCHANGE_TIMEOUT_MS (default 20 000)
 * fails with VT_ERR_DEVICE "RCCL exchange timed out on shard s ..." and poisons the handle. */
const char *vt_flat_exchange_note(const vt_flat *index);
/* Ranks of the shard communicator (ncclCommCount), 0 while none exists. */
int vt_flat_rccl_ranks(const vt_flat *index);
/* ResourceArc drop: frees HBM, streams, pinned staging. */
void vt_flat_free(vt_flat *index);

/* flat_insert/3, nifs.rs:259-271 -> FlatIndex::insert flat.rs:59-66. */
int vt_flat_insert(vt_flat *index, const char *id, size_t id_len,
                   const float *vector, size_t n);
/* flat_insert_many/2, nifs.rs:273-284 -> FlatIndex::insert_many flat.rs:69-85:
 * validates every row first (atomic), last duplicate id wins. */
int vt_flat_insert_many(vt_flat *index, size_t count, const char *ids,
                        const size_t *id_off, const float *values,
                        const size_t *value_off);
/* flat_delete/2, nifs.rs:286-295 -> FlatIndex::delete flat.rs:88-93.
 * Unknown id is a no-op; an emptied index forgets its dimension. */
int vt_flat_delete(vt_flat *index, const char *id, size_t id_len);
/* flat_search/3, nifs.rs:297-309 -> FlatIndex::search flat.rs:96-124.
 * limit == 0 returns an empty list before the query is validated. */
int vt_flat_search(vt_flat *index, const float *query, size_t n, size_t limit,
                   vt_hits **out);

/* Extension (SURVEY.md 8b "vt_flat_search_batch"): `nq` queries of `d` floats
 * each, stored back to back, against the same index.  out[i] receives query
 * i's hits -- identical (ids, order, raw bits) to nq vt_flat_search calls; the
 * first invalid query fails the whole call like its own flat_search would.
 * Dot-family metrics take the FP32-MFMA path: Q x D^T nominates candidates,
 * the exact kernel re-scores them, an error bound proves completeness. */
int vt_flat_search_batch(vt_flat *index, const float *queries, size_t nq, size_t d,
                         size_t limit, vt_hits **out);

size_t vt_flat_len(const vt_flat *index);
/* FlatIndex.dimension: -1 = None. */
long vt_flat_dimension(const vt_flat *index);
int vt_flat_metric(const vt_flat *index);
int vt_flat_set_reduce_order(vt_flat *index, int order);
/* Which matrix-core pass nominates the candidates of vt_flat_search_batch (and of coalesced
 * vt_flat_search calls): VT_NOMINATE_BF16 (default; operands rounded to bf16, the pass is
 * HBM-bound) or VT_NOMINATE_F32 (FP32 matrix cores).  Hits are identical bit for bit under
 * both: candidates are re-scored with the exact arithmetic and completeness is certified by
 * an error bound that knows the rounding.  Override for new indexes: VT_BATCH_NOMINATE=f32. */
#define VT_NOMINATE_F32 1
#define VT_NOMINATE_BF16 2
int vt_flat_set_batch_nominate(vt_flat *index, int mode);
int vt_flat_batch_nominate(const vt_flat *index);
/* The bf16 pass reads a bf16 SHADOW of the rows when the index keeps one: an image of every row,
 * rounded once (round to nearest even, what the pass itself would do on the fly) and laid out as
 * the matrix cores take their operands, kept beside the f32 slab -- half the bytes per pass, no
 * conversion in it.  It costs dimension * 2 bytes per row (half the slab again), is built by the
 * first batch that wants it, patched per mutated row like the other derived columns, and given
 * back at once when the rows themselves need the room (a growing slab never fails because of it).
 * The hits do not depend on it: same rounding, same bound, the exact kernel decides.
 *   VT_SHADOW_AUTO (default)  keep one when, after allocating it, at least a quarter of the
 *                             card's memory is still free
 *   VT_SHADOW_OFF             never (the pass streams the f32 rows and rounds in registers)
 * Override for new indexes: VT_BATCH_SHADOW=0.  vt_flat_batch_shadow reports shard 0's state. */
enum { VT_SHADOW_OFF = 0, VT_SHADOW_AUTO = 1 };
enum { VT_SHADOW_STATE_OFF = 0, VT_SHADOW_STATE_NONE = 1, VT_SHADOW_STATE_CURRENT = 2, VT_SHADOW_STATE_STALE = 3,
       VT_SHADOW_STATE_REFUSED = 4 };
int vt_flat_set_batch_shadow(vt_flat *index, int mode);
int vt_flat_batch_shadow(const vt_flat *index);   /* VT_SHADOW_STATE_*; -1: no index */
/* Opt-in (off by default; VT_SINGLE_NOMINATE=1 for new indexes): a lone vt_flat_search on an index whose bf16 shadow is
 * current is answered like a batch of one -- the bf16 pass over the shadow (half the bytes of the exact scan) nominates
 * a few hundred rows, the exact kernel re-scores them, the bound certifies that no other row can reach the top k; a
 * query the bound cannot certify takes the exact scan as before.  Same hits bit for bit, ~0.6 of the scan's time
 * (dot-family metrics, limit <= 256, ids ranked).  The default stays the exact scan of the f32 rows: what the
 * reference does (flat.rs:96-124), and what the headline benchmark measures. */
int vt_flat_set_single_nominate(vt_flat *index, int enabled);
int vt_flat_single_nominate(const vt_flat *index);
/* Order used by indexes created afterwards and by the stateless helpers. */
int vt_set_default_reduce_order(int order);

/* Bulk extensions (SURVEY.md 8b "vt_flat_load_matrix"): same semantics as
 * insert_many of `count` rows of equal length `d`, without the ragged layout.
 * _device takes a pointer to a row-major f32 [count][d] matrix already in this
 * device's HBM; finiteness is then checked by a device kernel. */
int vt_flat_load_matrix(vt_flat *index, size_t count, size_t d, const char *ids,
                        const size_t *id_off, const float *rows);
int vt_flat_load_device_matrix(vt_flat *index, size_t count, size_t d,
                               const char *ids, const size_t *id_off,
                               const void *device_rows);

/* quantized_search, lib/vettore/collection.ex:276-295: sign-bit Hamming
 * candidate pass (search.rs:76-92 over compress_sign_bits of every stored row,
 * distances.rs:413-437) followed by exact rerank (search.rs:38-73; cosine
 * reranks with the f64 `cosine`, search.rs:56-60). */
int vt_flat_quantized_search(vt_flat *index, const float *query, size_t n,
                             size_t candidates, size_t limit, vt_hits **out);
/* `nq` quantized searches (queries of `d` floats back to back), out[i] = query i's hits --
 * identical to nq vt_flat_quantized_search calls; groups of up to eight share ONE sweep of the
 * sign-bit matrix (candidates <= 256).  Concurrent vt_flat_quantized_search callers on one handle
 * are grouped the same way (they meet in the coalescer like plain searches do). */
int vt_flat_quantized_search_batch(vt_flat *index, const float *queries, size_t nq, size_t d,
                                   size_t candidates, size_t limit, vt_hits **out);

/* funnel_search, lib/vettore/collection.ex:245-260, :674-691: for every prefix
 * length in `stages` (1..dimensions, else "invalid prefix dimensions") keep the
 * `candidates` best rows by vector_top_k on that prefix (search.rs:38-73; a
 * cosine collection scores prefixes with the f64 `cosine`), the first stage over
 * the whole corpus, then exact rerank on the full vectors, top `limit`. */
int vt_flat_funnel_search(vt_flat *index, const float *query, size_t n,
                          const size_t *stages, size_t nstages, size_t candidates,
                          size_t limit, vt_hits **out);
/* `nq` funnel searches with one set of stages (queries of `d` floats back to back), out[i] = query
 * i's hits -- identical to nq vt_flat_funnel_search calls.  Groups of up to eight share ONE sweep of the
 * rows' first stages[0] coordinates (candidates <= 256, 16 384 rows or more): the f64 cosine on a cosine
 * collection, the metric's own f32 arithmetic on dot / L2 / manhattan / chebyshev ones (r04); under float
 * hamming / jaccard every query is a pass over the prefix of the non-zero-bit column (r04).  Anything else runs
 * query by query. */
int vt_flat_funnel_search_batch(vt_flat *index, const float *queries, size_t nq, size_t d,
                                const size_t *stages, size_t nstages, size_t candidates,
                                size_t limit, vt_hits **out);

/* hybrid_search with rerank: :exact, lib/vettore/collection.ex:325-345, :515-592:
 * the union (first occurrence wins) of the candidate sets of `ngen` generators,
 * exact rerank on the full vectors, top `limit`.  Generator i is
 *   VT_GEN_FUNNEL    stages[stage_off[i] .. stage_off[i+1]) prefix passes keeping candidates[i] rows
 *   VT_GEN_QUANTIZED sign-bit Hamming top candidates[i]
 *   VT_GEN_SEARCH    the index's own search with limit = candidates[i]            */
enum { VT_GEN_FUNNEL = 0, VT_GEN_QUANTIZED = 1, VT_GEN_SEARCH = 2 };
int vt_flat_hybrid_search(vt_flat *index, const float *query, size_t n, const int *kinds,
                          const size_t *candidates, const size_t *stage_off,
                          const size_t *stages, size_t ngen, size_t limit, vt_hits **out);

/* ---- row-sharded search across GPUs, device-side exchange -----
 * A shard's candidates are u64 keys (rank key << 32 | id_rank); they compare
 * across shards iff every shard's id_rank column comes from ONE ordering of all
 * ids.  vt_rank_ids computes that ordering (bytewise, like FlatHit::cmp,
 * flat.rs:34-40) for the concatenated ids of all shards; each shard installs its
 * slice with vt_flat_set_id_ranks (valid until its next mutation).
 *
 * Per query: vt_flat_search_begin enqueues upload + scan + select on the index's
 * stream and returns WITHOUT waiting; the shard's result lands in `device_block`
 * (device memory, >= 16 + limit * 16 bytes: {i32 status, u32 count, pad[2]} then
 * `limit` entries {u64 key, u32 row, f32 raw}).  The caller gathers the blocks of
 * all shards with one collective queued on vt_flat_stream(), then
 * vt_flat_merge_gathered merges `world` blocks on the device and waits once. */
int vt_rank_ids(const char *ids, const size_t *id_off, size_t count, uint32_t *out_rank);
int vt_flat_set_id_ranks(vt_flat *index, const uint32_t *ranks, size_t count);
void *vt_flat_stream(vt_flat *index); /* hipStream_t */
int vt_flat_search_begin(vt_flat *index, const float *query, size_t n, size_t limit,
                         void *device_block);
int vt_flat_merge_gathered(vt_flat *index, const void *device_blocks, size_t world,
                           size_t limit, size_t block_bytes, uint64_t *keys,
                           uint32_t *rows, float *raw, uint32_t *shard, size_t *count);

/* ------------------------------------------------- stateless NIF helpers
 * vector_top_k/5, nifs.rs:151-162 -> search.rs:38-73. */
int vt_vector_top_k(int device, size_t count, const char *ids,
                    const size_t *id_off, const float *values,
                    const size_t *value_off, const float *query, size_t nq,
                    int metric_code, size_t dimensions, size_t limit,
                    vt_hits **out);
/* binary_top_k/4, nifs.rs:164-175 -> search.rs:76-92. */
int vt_binary_top_k(int device, size_t count, const char *ids,
                    const size_t *id_off, const uint64_t *words,
                    const size_t *word_off, const uint64_t *query, size_t nq,
                    size_t dimensions, size_t limit, vt_hits **out);
/* normalize_l2/1, nifs.rs:107-111 -> distances.rs:350-361 (rows of a
 * [count][d] matrix; count == 1 is the NIF call). */
int vt_normalize_l2(int device, size_t count, size_t d, const float *in,
                    float *out);
/* compress_sign_bits/1, nifs.rs:125-129 -> distances.rs:413-423.
 * out has count * ((d + 63) / 64) words. */
int vt_compress_sign_bits(int device, size_t count, size_t d, const float *in,
                          uint64_t *out);

/* ------------------------------------------------------------ profiling
 * Device-side timing of the dominant kernels with HIP events on the stream the
 * kernels are launched on (bench.py's roofline figure). */
typedef struct vt_profile {
  uint64_t scan_launches;   /* flat scan + fused top-k kernel launches */
  double scan_ms;           /* sum of their durations (hipEventElapsedTime) */
  uint64_t scan_rows;       /* rows scanned by those launches */
  uint64_t scan_bytes;      /* algorithmic bytes: rows * d * 4 */
  uint64_t hamming_launches; /* packed-bit passes: quantized candidates; flat searches under float hamming / jaccard */
  double hamming_ms;
  uint64_t hamming_bytes;   /* rows * ceil(d/64) * 8 per launch (one launch may serve up to 8 queries) */
  uint64_t merge_launches;
  double merge_ms;
  uint64_t batch_launches;  /* MFMA candidate passes (one per <= 256 queries) */
  double batch_ms;
  double batch_flops;       /* 2 * rows * padded queries * padded dims per pass */
  uint64_t batch_queries;
  uint64_t batch_fallbacks; /* queries the bound could not certify (re-run singly) */
  uint64_t prefix_launches; /* funnel stages over ALL rows: f64 cosine prefix scans, K1 / K1p prefix sweeps, K4 passes over the bit column's prefix */
  double prefix_ms;
  uint64_t prefix_bytes;    /* rows * prefix dimensions * 4 (rows * ceil(prefix / 64) * 8 from the bit column) */
  /* K2b: candidate passes with bf16 operands (HBM-bound; batch_* above count the FP32 passes) */
  uint64_t nominate_launches;
  double nominate_ms;
  uint64_t nominate_bytes;        /* algorithmic bytes: what a pass reads -- rows * d * 4 from the f32 rows, rows * d * 2 from the bf16 shadow */
  double nominate_flops;          /* 2 * rows * 256 * padded dims per pass */
  uint64_t nominate_queries;
  uint64_t nominate_second_passes; /* passes re-run with thresholds from a first pass's exact hits */
  uint64_t nominate_candidates;   /* rows handed to the exact rescoring, summed over queries */
  uint64_t hamming_queries;       /* queries served by grouped passes over a bit column: quantized groups, float hamming / jaccard batches (0 for single-query passes) */
  uint64_t hybrid_device_chains;  /* hybrid searches whose generators, union and rerank ran as one device chain (one host wait) */
  uint64_t prefix_queries;        /* queries served by grouped prefix scans (0 for single-query funnel searches) */
  /* r04 (VT_ABI_VERSION 3) */
  uint64_t nominate_shadow_launches; /* of nominate_launches: the passes that read the bf16 shadow (K2s) */
  uint64_t shadow_builds;            /* whole-image builds of the bf16 shadow */
  double shadow_build_ms;
  uint64_t shadow_patched_rows;      /* rows re-rounded in place after mutations */
  uint64_t sweep_queries;            /* plain searches of a batch served by K1p sweeps (their sweeps count as scan_launches) */
} vt_profile;
int vt_flat_set_profiling(vt_flat *index, int enabled);
int vt_flat_get_profile(vt_flat *index, vt_profile *out, int reset);
/* The same for a caller that says how large ITS vt_profile is: at most `out_bytes` are written
 * (a binding built against an older header keeps working). */
int vt_flat_get_profile_sized(vt_flat *index, void *out, size_t out_bytes, int reset);

#ifdef __cplusplus
}
#endif
#endif
