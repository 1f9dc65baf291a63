// vt_device.h -- launch interface between the host index (vt_index.cpp) and the
// gfx950 kernels (vt_*.hip).  Internal; the public boundary is
// include/vettore_flat.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vt {

// Candidate key of a row for one query:
//   key = orderable(rank_value(metric, raw)) << 32 | id_rank
// Ascending u64 order of `key` == the reference's (rank.total_cmp, id bytes)
// order (flat.rs:34-40): `id_rank` is order-isomorphic to the bytewise order of
// the row's id among the rows of this index.
struct Payload {
  uint32_t row;
  float raw;
};
struct Entry {
  uint64_t key;
  uint32_t row;
  float raw;
};
static_assert(sizeof(Entry) == 16 && sizeof(Payload) == 8, "layout");

constexpr uint64_t kEmptyKey = ~0ull;
constexpr int kTileRows = 32;     // rows per wave tile in the scan kernel
constexpr int kWavesPerBlock = 4;
constexpr int kMaxFusedK = 256;   // largest k one scan pass selects (320-slot wave buffer)
constexpr int kSmallK = 64;       // k <= kSmallK uses the 128-slot wave buffer
constexpr uint32_t kRowAlign = 64;  // slab row stride is a multiple of 64 floats (256 B)

// What the select kernel hands back (lives in pinned host memory: the kernel
// writes it through the PCIe mapping, the host reads it after the stream sync).
struct ResultBlock {
  int status;
  uint32_t count;
  uint32_t pad[2];
  Entry e[kMaxFusedK];
};

__host__ __device__ inline uint32_t padded_dim(uint32_t d) { return (d + kRowAlign - 1) / kRowAlign * kRowAlign; }

struct ScanArgs {
  const float *X;          // row-major rows, `stride` floats apart, 256-B aligned
  size_t stride;           // floats between rows (multiple of 64, >= padded_dim(d))
  const float *q;          // query padded with zeros to padded_dim(d) floats (device)
  const uint32_t *id_rank; // per row (indexed like rows of X); null => row index
  const uint32_t *gather;  // optional: scan rows X[gather[i * gather_stride]] for i < n
  uint32_t gather_stride;  // in uint32 units
  uint32_t n;              // rows to scan
  uint32_t d;              // dimensions used (prefix length)
  int metric;              // VT_* metric code
  int order;               // VT_ORDER_*
  uint32_t k;              // 1..kMaxFusedK
  uint64_t lo_key;         // keep only keys > lo_key when has_lo
  int has_lo;
  uint32_t q_nonzero;      // count of query coordinates != 0 (Jaccard)
  uint64_t *part_keys;     // [grid_blocks][k]: one merged list per block
  Payload *part_pay;       // [grid_blocks][k]
  int *status;             // device int; atomicMax'ed to VT_ERR_OVERFLOW on "metric overflow"
  // Batch mode (gathered scans only, grid.y = queries): query b = blockIdx.y uses
  // q + b * padded_dim(d), gather + b * batch_cap * gather_stride, scans
  // min(batch_counts[b], batch_cap) rows and writes list (b * grid.x + blockIdx.x).
  const uint32_t *batch_counts;
  uint32_t batch_cap;
  uint32_t batch_qstride;       // floats between the queries of a batch (0: padded_dim(d))
  uint32_t batch_gather_stride; // u32 words between the row lists of a batch (0: batch_cap * gather_stride)
  uint32_t tile_rows;      // 0 / 32 (default), 16 or 8: rows per wave tile, see scan_tile_rows()
  // Key-column mode (limits above kMaxFusedK): instead of keeping the k best, every scanned
  // position i writes key_out[i] (kEmptyKey if the row is excluded) and, if given, pay_out[i].
  uint64_t *key_out;
  Payload *pay_out;
};

// Rows per wave tile for a scan of n rows of dimension d on `resident_waves` waves:
// 32 when there is plenty of work per wave; 16 or 8 when there are few tiles per
// wave (a 32-row tile of 768 floats is ~25 us of one wave's load latency, so small
// corpora spread over more waves).  ntiles = ceil(n / result).
uint32_t scan_tile_rows(uint32_t n, uint32_t d, uint32_t resident_waves);

// LDS bytes per block the scan kernel needs for dimension d and list size k (0 = unsupported).
size_t scan_lds_bytes(uint32_t d, uint32_t k);
// LDS bytes per block of the hamming kernel for list size k.
size_t hamming_lds_bytes(uint32_t k);
// Partial lists a launch with `blocks` blocks produces (one per block).
inline uint32_t scan_lists(uint32_t blocks) { return blocks; }
hipError_t launch_scan(const ScanArgs &a, uint32_t blocks, hipStream_t s);

// K1m (vt_scan_multi.hip): the same scan for up to kMultiMaxQueries queries in ONE sweep of
// the corpus, exact arithmetic, any metric; contiguous rows only.  Block b leaves query q's
// list at part_keys/part_pay[((first_query + q) * grid_blocks + b) * k ..) -- the layout
// launch_batch_select reads.
constexpr uint32_t kMultiMaxQueries = 8;
struct MultiScanArgs {
  const float *X;
  size_t stride;
  const float *Q;           // [nq][ld] queries, zero padded (device)
  const uint32_t *id_rank;
  uint32_t n, d, ld;        // ld = padded_dim(d)
  int metric, order;
  uint32_t k;               // <= scan_multi_max_k(nq)
  uint32_t nq;              // queries of this launch, 1..kMultiMaxQueries
  uint32_t first_query;     // index of Q[0] among the lists of the whole batch
  uint32_t dbg;             // timing experiments (builds with -DVT_MULTI_TIMING_EXPERIMENTS only; VT_MQ_DBG): see the kernel
  uint32_t q_nonzero[kMultiMaxQueries];
  uint64_t *part_keys;
  Payload *part_pay;
  int *status;
};
uint32_t scan_multi_max_k(uint32_t nq);       // 64 for up to 4 queries per sweep, 32 for up to 8
uint32_t scan_multi_tile_rows(uint32_t nq);   // rows per wave tile (grid sizing)
size_t scan_multi_lds_bytes(uint32_t d, uint32_t k, int metric);
int scan_multi_blocks_per_cu(uint32_t d, uint32_t k, int metric);  // 3 for the slim build (d % 64 == 0, k <= 16, dot/L2/L1/Linf), else 2
hipError_t launch_scan_multi(const MultiScanArgs &a, uint32_t blocks, hipStream_t s);
// Batch mode: `blocks` per query x `nq` queries.
hipError_t launch_scan_batch(const ScanArgs &a, uint32_t blocks, uint32_t nq, hipStream_t s);

// K3: selects the k smallest of keys[0..m) (kEmptyKey ignored) by radix select,
// writes them sorted ascending with their payloads to out->e, out->count;
// moves *dev_status into out->status and clears it for the next query.
// Keys <= lo_key are ignored when has_lo (multi-pass selection of k > kMaxFusedK).
// Lists of >= kSelTwoLevelMin keys are selected in two levels (kSelGroups blocks on
// slices, then one block); scratch_keys/scratch_pay hold kSelGroups * k entries.
// Exact k-th smallest of a key column by three 11-bit radix passes over the top 33 bits
// (the whole rank key and the top id-rank bit) -- or six passes over all 64 bits (RadixArgs.passes)
// --, all decisions taken on the device:
// launch_radix_pass for pass = 0, 1, 2 (.. 5) (hist zeroed beforehand), then launch_radix_collect
// appends every key below the resolved 33-bit prefix, and the keys sharing it, to a list
// (positions as payload rows); the caller selects the k best of that list.  A list overflow
// raises kStatusRetry in *status.
constexpr uint32_t kRadixBins = 2048;
struct RadixArgs {
  const uint64_t *keys;
  uint32_t n;
  uint32_t k;
  uint32_t *hist;        // [3][kRadixBins]
  uint32_t *list_count;  // cleared by pass 0
  uint64_t *list_keys;   // [cap]
  Payload *list_pay;     // [cap]: row = position in `keys`
  uint32_t cap;
  int *status;
  const Payload *pay_col;  // optional: per-position payload whose raw value travels with a collected key
  // 3 (default when 0): the threshold is the k-th key's top 33 bits (its whole rank and the top
  // id-rank bit), keys sharing them are all collected.  6: all 64 bits -- with unique keys the
  // collect pass leaves exactly the k smallest, however many rows tie in their rank
  // (hist then holds 6 * kRadixBins counters).
  int passes;
};
hipError_t launch_radix_pass(const RadixArgs &a, int pass, uint32_t blocks, hipStream_t s);
hipError_t launch_radix_collect(const RadixArgs &a, uint32_t blocks, hipStream_t s);

// A whole list of <= kSelListMax (key, payload) pairs in ascending key order into host-mapped
// memory: `out` receives the live entries, `head` their count and the scan status word.
struct BigResultHeader {
  int status;
  uint32_t count;
};
hipError_t launch_sort_list(const uint64_t *keys, const Payload *pay, uint32_t m, int *dev_status, BigResultHeader *head,
                            Entry *out, hipStream_t s);

// The k smallest of a key list as a device-resident, unsorted list (k <= kSelListMax): the
// candidate set of a following stage whose own ordering does not depend on this one.
constexpr uint32_t kSelListMax = 4096;
hipError_t launch_select_list(const uint64_t *keys, const Payload *pay, uint32_t m, const uint32_t *m_dev, uint32_t k,
                              uint64_t *out_keys, Payload *out_pay, hipStream_t s);
// One radix select per query in ONE launch (grid.y = queries): query y's list is keys/pay[y * m ..
// (y + 1) * m), its winners land, sorted, in the block at out + y * out_stride bytes
// ({i32 status = 0, u32 count, pad[2]} + k entries).
hipError_t launch_select_queries(const uint64_t *keys, const Payload *pay, uint32_t nq, uint32_t m, uint32_t k, void *out,
                                 uint32_t out_stride, hipStream_t s);
constexpr uint32_t kSelGroups = 16;
constexpr uint32_t kSelTwoLevelMin = 16384;
hipError_t launch_select(const uint64_t *keys, const Payload *pay, uint32_t m, uint32_t k, uint64_t lo_key, int has_lo,
                         int *dev_status, ResultBlock *out, uint64_t *scratch_keys, Payload *scratch_pay,
                         hipStream_t s, const uint32_t *m_dev = nullptr);

// K4's bit-matrix layout: per tile of 64 rows, word pair j of all 64 rows is
// contiguous -- [tile][pair][row % 64][2] u64 (an odd word count is padded with a
// zero word).  Index of word `w` of row `r`:
__host__ __device__ inline size_t hamming_word_index(uint32_t r, uint32_t w, uint32_t pairs) {
  return ((size_t)(r / 64) * pairs + (w >> 1)) * 128 + (size_t)(r % 64) * 2 + (w & 1);
}
// u64 words the tiled matrix of n rows occupies.
inline size_t hamming_matrix_words(uint32_t n, uint32_t words) {
  return (size_t)((n + 63) / 64) * ((words + 1) / 2) * 128;
}

struct HammingArgs {
  const uint64_t *bits;    // tiled layout above, hamming_matrix_words(n, words) words, zero-filled padding
  const uint64_t *qbits;   // [words] plain (device)
  const uint32_t *id_rank; // per row or null
  uint32_t n, words, pairs, d, k;  // pairs = (words + 1) / 2
  uint64_t lo_key;
  int has_lo;
  uint64_t *part_keys;
  Payload *part_pay;
  int jaccard;  // the bits are non-zero patterns and the score is distances.rs:327-347's (else the differing bits)
  // A PREFIX of the rows' bits (funnel stages under float hamming / jaccard, vector_top_k on the first d
  // coordinates, search.rs:38-73): words / pairs / d describe the prefix, tile_pairs the word pairs a tile of
  // the matrix really holds (0: the same as `pairs`).  Words behind the prefix are never counted.
  uint32_t tile_pairs;
};
hipError_t launch_hamming(const HammingArgs &a, uint32_t blocks, hipStream_t s);

// K4p (vt_kernels.hip): K4's pattern mode (non-zero bits; float hamming / jaccard scores) for up to
// kPatternMultiMax queries in one sweep of the column; k <= kSmallK; unsorted lists of k per
// (query, block) at part_keys / part_pay + ((first_query + q) * blocks + block) * k.
constexpr uint32_t kPatternMultiMax = 8;
struct PatternMultiArgs {
  const uint64_t *bits;     // the non-zero-bit column, K4's tiled layout
  const uint64_t *qbits;    // [nq][2 * pairs] (device): each query's words, an odd count padded with a zero word
  const uint32_t *id_rank;  // per row or null
  uint32_t n, words, pairs, d, k, nq, first_query;
  int jaccard;
  uint64_t *part_keys;
  Payload *part_pay;
};
size_t pattern_multi_lds_bytes();
bool pattern_multi_supports(uint32_t pairs);  // word-pair counts with an unrolled build (d up to 2 048 in steps)
hipError_t launch_pattern_multi(const PatternMultiArgs &a, uint32_t blocks, hipStream_t s);

// K4h (vt_kernels.hip): distance column + histogram, then threshold collect.
constexpr uint32_t kHammingHistMaxDim = 8191;  // (d + 1) u32 bins must fit comfortably in LDS
constexpr int kStatusRetry = 100;              // internal: the tie list overflowed, take the K4 path
struct HammingHistArgs {
  const uint64_t *bits;   // tiled layout, as for K4
  const uint64_t *qbits;  // [words]
  uint32_t n, words, pairs, d;
  uint16_t *dist;         // [n] out
  uint32_t *hist;         // [d + 1] this query's histogram (zero on entry)
  uint32_t *list_count;   // cleared here for the collect pass
};
struct HammingCollectArgs {
  const uint16_t *dist;
  const uint32_t *id_rank;
  uint32_t n, d, k;
  const uint32_t *hist;
  uint32_t *hist_next;    // the other histogram, cleared for the next query (null: nothing to clear)
  uint32_t *list_count;
  uint64_t *keys;         // [cap]
  Payload *pay;           // [cap]
  uint32_t cap;
  int *status;
  // launch_hamming_collect_multi (the queries of a group in one launch): `dist` is the
  // interleaved column dist[row][8] (u16), query q's histogram starts at hist + q * hist_stride,
  // its count is list_count[q], its list keys / pay + q * cap
  uint32_t dist_stride, hist_stride;
};
// K4h for up to kHammingMultiMax queries in ONE sweep of the bit matrix (concurrent / batched
// quantized searches): row r's eight distances land in dist[8 r .. 8 r + 8) (one 16-byte store),
// query q's histogram in hist + q * hist_stride (zeroed beforehand); list_count[0..8) is cleared
// for the collect pass.  qbits: [nq][2 * pairs] words, padding bits and the padding word clear.
constexpr uint32_t kHammingMultiMax = 8;
struct HammingMultiArgs {
  const uint64_t *bits;   // tiled layout, as for K4
  const uint64_t *qbits;  // [nq][words]
  uint32_t n, words, pairs, d, nq;
  uint16_t *dist;         // [n][8], 16-byte aligned
  uint32_t dist_stride;   // (unused: the column is interleaved)
  uint32_t *hist;
  uint32_t hist_stride;   // in u32, >= d + 1
  uint32_t *list_count;   // [nq]
};
size_t hamming_multi_lds_bytes(uint32_t d, uint32_t words, uint32_t nq);
hipError_t launch_hamming_dist_multi(const HammingMultiArgs &a, uint32_t blocks, hipStream_t s);
// the collect pass for the nq queries of a group in one sweep of the interleaved distance column
hipError_t launch_hamming_collect_multi(const HammingCollectArgs &a, uint32_t blocks, uint32_t nq, hipStream_t s);
// K3 for nq lists whose lengths were decided on the device: list y = keys / pay + y * m_stride,
// m_dev[y] entries; winners (sorted) to the block at out + y * out_stride bytes.
hipError_t launch_select_lists(const uint64_t *keys, const Payload *pay, uint32_t nq, uint32_t m_stride, const uint32_t *m_dev,
                               uint32_t k, void *out, uint32_t out_stride, hipStream_t s, bool spread = false);
size_t hamming_hist_lds_bytes(uint32_t d);
hipError_t launch_hamming_dist(const HammingHistArgs &a, uint32_t blocks, hipStream_t s);
hipError_t launch_hamming_collect(const HammingCollectArgs &a, uint32_t blocks, hipStream_t s);

// rows[n][stride] (first d columns) -> sign bits, bit j%64 of word j/64 set iff
// v[j] >= 0.0 (distances.rs:413-423).  tiled: K4's layout, else plain [n][words].
// nonzero: the bit says v[j] != 0.0 instead (the "truthiness" float hamming / jaccard compare,
// distances.rs:319-347).
hipError_t launch_sign_pack(const float *rows, size_t stride, uint32_t n, uint32_t d, uint64_t *bits, int tiled,
                            hipStream_t s, int nonzero = 0);

// *flag |= 1 if any of the first d columns of any row is non-finite.
hipError_t launch_check_finite(const float *rows, size_t stride, uint32_t n, uint32_t d, int *flag, hipStream_t s);

// dst[n][dst_stride] <- src[n][d], columns d..dst_stride-1 zero-filled.
hipError_t launch_pad_rows(const float *src, uint32_t n, uint32_t d, float *dst, size_t dst_stride, hipStream_t s);

// dst row map[2i + 1] <- src row map[2i] for i < count (map on the device; src rows are d floats,
// dst rows dst_stride floats, zero padded).
// A trickle of host rows lands from a pinned slot (layout in vt_kernels.hip: rows, their slab rows, id ranks): one launch.
hipError_t launch_land_rows(const float *stage_dev, uint32_t count, uint32_t ld, float *X, uint32_t *rank_col, uint32_t rank_first,
                            uint32_t nranks, hipStream_t s);
// Swap-delete on the slab: row `last` moves into row r (with its rank when rank_col is given), row `last` is zeroed.
hipError_t launch_swap_delete(float *X, uint32_t ld, uint32_t r, uint32_t last, uint32_t *rank_col, hipStream_t s);
hipError_t launch_gather_rows(const float *src, uint32_t d, const uint32_t *map, uint32_t count, float *dst,
                              size_t dst_stride, hipStream_t s);

// K5 / row norms for a device list of rows (derived data of mutated rows patched in place).
hipError_t launch_sign_pack_rows(const float *rows, size_t stride, const uint32_t *list, uint32_t count, uint32_t d,
                                 uint64_t *bits, hipStream_t s, int nonzero = 0);
hipError_t launch_row_sqnorms_rows(const float *X, size_t stride, const uint32_t *list, uint32_t count, uint32_t d,
                                   float *xnorm2, unsigned long long *out_bits, hipStream_t s);

// dst[pairs[2i]] = pairs[2i + 1] for i < n (pairs on the device).
hipError_t launch_scatter_u32(const uint32_t *pairs, uint32_t n, uint32_t *dst, hipStream_t s);

// Exact rerank for Metric::Cosine (search.rs:56-60 -> distances.rs:160-177):
// raw = clamp(f64_dot(q,x) / (sqrt(f64_dot(q,q)) * sqrt(f64_dot(x,x)))) as f32,
// one candidate per lane, sequential f64 sums.  Emits key/payload per candidate.
struct CosineRerankArgs {
  const float *X;
  size_t stride;
  const float *q;          // device, >= d floats
  const uint32_t *id_rank; // per row or null
  const uint32_t *gather;  // candidate rows (null => rows 0..n-1)
  uint32_t gather_stride;
  uint32_t n;              // candidates
  uint32_t d;
  uint64_t *out_keys;      // [n]
  Payload *out_pay;        // [n]
  int *status;
  // several queries in one launch (launch_cosine_rerank_batch, grid.y = queries): query y uses
  // q + y * q_stride, gather + y * gather_qstride and writes to out_keys / out_pay + y * n
  uint32_t q_stride, gather_qstride;
  // when set (single-query launches): only the first min(n, *n_dev) candidates exist, the other
  // slots get the empty key -- a candidate list whose length only the device knows
  const uint32_t *n_dev;
};
hipError_t launch_cosine_rerank(const CosineRerankArgs &a, hipStream_t s);
hipError_t launch_cosine_rerank_batch(const CosineRerankArgs &a, uint32_t nq, hipStream_t s);

// Cross-shard merge on the device: `blocks` is `world` ResultBlock prefixes
// (16-B header + k entries each, `block_bytes` apart) as gathered from the shards;
// writes the k smallest keys overall, sorted, to out->e / out->count, the shard of
// each winner to out_shard, and the OR of the shards' status words to out->status.
hipError_t launch_merge_blocks(const void *blocks, uint32_t world, uint32_t k, uint32_t block_bytes, ResultBlock *out,
                               uint32_t *out_shard, hipStream_t s);

// ---- K2: query batches on the FP32 matrix cores (vt_batch.hip) ----------------
struct BatchCand {
  float score;   // MFMA (approximate-order) dot product
  uint32_t row;
};
struct BatchScoreArgs {
  const float *X;         // slab
  size_t stride;
  const float *Q;         // [nq_pad][ld] queries, zero padded (device)
  uint32_t ld;            // padded_dim(d)
  uint32_t nq_pad;        // 32, 64, 128 or 256
  uint32_t n;             // rows covered by this launch (pass 0: sample rows)
  uint32_t n_total;       // rows in the index
  uint32_t sample_stride; // pass 0: tile i of the launch is row tile i * sample_stride
  float *sample;          // pass 0: [nq_pad][sample_rows] dense scores
  uint32_t sample_rows;
  const float *tau;       // pass 1: per-query candidate threshold
  BatchCand *cand;        // pass 1: [nq_pad][cand_cap]
  uint32_t *cand_count;   // pass 1: [nq_pad], may exceed cand_cap (overflow)
  uint32_t cand_cap;
  const float *xnorm2;    // null: score = q.x; else score = 2 q.x - xnorm2[row] (= |q|^2 - |q - x|^2)
  uint32_t debug;         // VT_BATCH_DEBUG timing experiments (results invalid when non-zero)
  const void *Qimage;     // K2b / K2s: the queries rounded to bf16, in fragment order (launch_batch_q_image / _q_image16)
  const void *Xshadow;    // K2s only: the rows rounded to bf16, in fragment order (shadow_index; launch_shadow_build)
  uint32_t stages;        // K2s only: depth of the LDS ring for this launch -- 4 or 5; 0: the library's default (VT_SHADOW_STAGES)
};
uint32_t batch_rows_per_block(uint32_t nq_pad);
hipError_t launch_batch_scores(const BatchScoreArgs &a, bool dense, uint32_t blocks, hipStream_t s);
// K2b (vt_batch_bf16.hip): the same two passes with both operands rounded to bf16 on the way
// into v_mfma_f32_32x32x16_bf16 -- HBM-bound instead of FP32-MFMA-bound.  Always 256 query
// columns (nq_pad = 64, 128 or 256 -- batch_bf16_pad --, the padding ones zero); a.Qimage from launch_batch_q_image.
uint32_t batch_bf16_rows_per_block();
size_t batch_bf16_image_bytes(uint32_t ld);
uint32_t batch_bf16_pad(uint32_t nq);  // columns a batch of nq <= 256 queries is padded to: 64, 128 or 256
hipError_t launch_batch_q_image(const float *Q, uint32_t ld, uint32_t nq_pad, void *image, hipStream_t s);
hipError_t launch_batch_scores_bf16(const BatchScoreArgs &a, bool dense, uint32_t blocks, hipStream_t s);
// K2s (vt_batch_shadow.hip): K2b fed from a bf16 image of the rows kept beside the slab -- half the
// HBM bytes, no conversion in the pass, both operands through LDS as whole fragments of
// v_mfma_f32_16x16x32_bf16.  Same rounding as K2b's (round to nearest even), so the same bound applies.
// Element (row, col) of the image of rows `ld` floats long (ld a multiple of 64) sits at
// [row / 16][col / 32][(col / 8) % 4][row % 16][col % 8]: 1 KiB per (16 rows, 32 columns), lane
// 16 g + r of a wave holding row r, k = 8 g .. 8 g + 7 -- one A operand.
__host__ __device__ inline size_t shadow_index(uint32_t row, uint32_t col, uint32_t ld) {
  return ((size_t)(row >> 4) * (ld >> 5) + (col >> 5)) * 512 + ((col >> 3) & 3u) * 128 + (row & 15u) * 8 + (col & 7u);
}
uint32_t batch_shadow_rows_per_block();
size_t shadow_elems(uint32_t rows, uint32_t ld);   // bf16 elements of an image of `rows` rows (padded to whole block tiles)
size_t batch_shadow_image_bytes(uint32_t ld);      // the query image of up to 256 queries
// rows [0, rows_src) of X -> image rows [0, rows_img) (rows_img a multiple of 256; rows >= rows_src zero)
hipError_t launch_shadow_build(const float *X, size_t stride, uint32_t rows_src, uint32_t rows_img, uint32_t ld, void *img,
                               hipStream_t s);
// the rows of a device list (mutated rows patched in place)
hipError_t launch_shadow_rows(const float *X, size_t stride, const uint32_t *list, uint32_t count, uint32_t ld, void *img,
                              hipStream_t s);
hipError_t launch_batch_q_image16(const float *Q, uint32_t ld, uint32_t nq_pad, void *image, hipStream_t s);
hipError_t launch_batch_scores_shadow(const BatchScoreArgs &a, bool dense, uint32_t blocks, hipStream_t s);
// the sample pass in its r05 form: per query the best score of every 64-row group of the sampled tiles --
// a.sample is [nq_pad][a.sample_rows], a.sample_rows = batch_shadow_sample_groups(tiles of the launch)
hipError_t launch_batch_sample_maxima_shadow(const BatchScoreArgs &a, uint32_t blocks, hipStream_t s);
uint32_t batch_shadow_sample_groups(uint32_t sample_tiles);
// tau[b] for the nq_real real queries; +inf for the padding columns b >= nq_real.
// (K2s since r05: the threshold from the sample's group maxima, [nq][groups] with groups <= 1 024)
hipError_t launch_sample_tau_groups(const float *maxima, uint32_t groups, uint32_t nq, uint32_t nq_real, uint32_t rank, float *tau,
                                    hipStream_t s);
hipError_t launch_sample_tau(const float *sample, uint32_t sample_rows, uint32_t nq, uint32_t nq_real, uint32_t rank,
                             float *tau, hipStream_t s);
// xnorm2[i] = (f32) sum_j x_ij^2 (f64 accumulation); *out_bits = bit pattern of
// the f64 maximum over the rows (zero it first).
hipError_t launch_row_sqnorms(const float *X, size_t stride, uint32_t n, uint32_t d, float *xnorm2,
                              unsigned long long *out_bits, hipStream_t s);
// Block b: the k smallest of keys[b][0..m) sorted ascending -> out[b][0..k), out_count[b].
// (`ex`: see batch_select_kernel -- the host-side numbers of a query batch leave with the lists)
struct BatchExport {
  const uint32_t *cand_count;  // [nq] -> cand_count_out
  uint32_t *cand_count_out;
  const float *tau;            // [nq] -> tau_out (null: not wanted)
  float *tau_out;
  int *status;                 // -> *status_out, then 0
  int *status_out;
};
hipError_t launch_batch_select(const uint64_t *keys, const Payload *pay, uint32_t nq, uint32_t m, uint32_t k, Entry *out,
                               uint32_t *out_count, hipStream_t s, const BatchExport *ex = nullptr);

// K6b: exact f64 cosine (distances.rs:160-177) of the first d coordinates of
// EVERY row against the query, fused top-k -- stage 1 of funnel_search on a
// cosine collection (collection.ex:245-260 -> search.rs:56-60).
struct CosineScanArgs {
  const float *X;
  size_t stride;
  const float *q;          // device, padded with zeros to padded_dim(d)
  double qq;               // f64_dot(q, q) over the first d coordinates (sequential, host)
  const uint32_t *id_rank;
  uint32_t n, d, k;
  uint64_t lo_key;
  int has_lo;
  uint64_t *part_keys;     // [grid_blocks][k]
  Payload *part_pay;
  int *status;
  uint64_t *key_out;       // key-column mode, as in ScanArgs
};
size_t cosine_scan_lds_bytes(uint32_t d, uint32_t k);
hipError_t launch_cosine_scan(const CosineScanArgs &a, uint32_t blocks, hipStream_t s);

// K6b for up to kCosineMultiMax queries in ONE sweep of the rows' prefixes (funnel_search's stage 1,
// collection.ex:245-260 -> search.rs:56-60, for several callers at once): every lane owns a row and
// carries x.x once and q.x per query as sequential f64 chains -- the values are the single kernel's,
// bit for bit.  No per-wave top-k buffers (eight of them would not fit beside the row panels):
//   dense mode (`sample` set): the scores of every `sample_stride`-th tile of 64 rows go to
//     sample[q * sample_rows + i] -- launch_sample_tau turns them into one threshold per query;
//   sweep mode: every (query, row) with raw >= tau[q] is appended to the query's list as
//     (key, {row, raw}) -- key as in the single kernel -- cand_count[q] counting ALL of them;
//     the host checks k <= count <= cap (else that query takes the single path) and
//     launch_select_lists cuts each list to its k best.
constexpr uint32_t kCosineMultiMax = 8;
struct CosineScanMultiArgs {
  const float *X;
  size_t stride;
  const double *Qd;          // [kCosineMultiMax][padded_dim(d)] device: the queries' prefixes as f64 (exact), unused rows zero
  double qq[kCosineMultiMax];  // f64_dot(q, q) over the first d coordinates (sequential, host)
  const uint32_t *id_rank;
  uint32_t n, d, nq;
  float *sample;             // dense mode when set
  uint32_t sample_stride, sample_rows;
  // (r05) dense mode files only the BEST score of every sampled 64-row tile: sample[q * sample_rows + tile], with
  // sample_rows = the launch's tiles; the threshold then comes from launch_sample_tau_groups (the K2s sample's scheme)
  uint32_t sample_maxima;
  const float *tau;          // [nq] sweep mode
  uint64_t *cand_keys;       // [nq][cand_cap]
  Payload *cand_pay;
  uint32_t *cand_count;      // [nq], zeroed by the caller
  uint32_t cand_cap;
  int *status;
};
size_t cosine_scan_multi_lds_bytes();
hipError_t launch_cosine_scan_multi(const CosineScanMultiArgs &a, uint32_t blocks, hipStream_t s);

// K1p: the index's own metric (K1's arithmetic: eight separately rounded products per chunk, the horizontal add in
// the lane order `order`, acc += chunk sum, the scalar tail -- distances.rs:197-262) over the first d coordinates of
// EVERY row for up to kPrefixMultiMax queries in ONE sweep of the prefixes: stage 1 of funnel_search on an L2 / dot /
// L1 / Linf collection (collection.ex:245-260 -> search.rs:56-60) for several callers at once.  Lane r walks row r
// through a 64 x 64-float LDS panel like K6b; the queries come through the scalar cache.  Modes as K6b's:
//   dense (`sample` set): -rank_value of every sample_stride-th tile's rows to sample[q * sample_rows + i]
//     (larger = better: launch_sample_tau's order);
//   sweep: every (query, row) with -rank_value >= tau[q] is appended to the query's list as (key, {row, raw}),
//     key as in K1; cand_count[q] counts all of them.
constexpr uint32_t kPrefixMultiMax = 8;
struct PrefixMultiArgs {
  const float *X;
  size_t stride;
  const float *Q;            // [kPrefixMultiMax][q_stride] f32 queries (device; unused rows readable), q_stride % 8 == 0
  uint32_t q_stride;
  const uint32_t *id_rank;
  uint32_t n, d, nq;
  int metric, order;         // metrics of the dot / L2 / L1 / Linf families (not cosine, not the pattern metrics)
  float *sample;
  uint32_t sample_stride, sample_rows;
  uint32_t sample_maxima;    // as in CosineScanMultiArgs
  const float *tau;
  uint64_t *cand_keys;       // [nq][cand_cap]
  Payload *cand_pay;
  uint32_t *cand_count;      // [nq], zeroed by the caller
  uint32_t cand_cap;
  int *status;
};
bool prefix_multi_supports(int metric);
size_t prefix_multi_lds_bytes();
int prefix_multi_blocks_per_cu();  // resident blocks per CU the launch is sized for (2)
hipError_t launch_prefix_multi(const PrefixMultiArgs &a, uint32_t blocks, hipStream_t s);

// Diagnostic (vt_device_read_peak): one pass of the bare LDS-DMA read stream over the whole 384-KiB tiles of
// `buf` (read_peak_bytes(bytes) of it), `blocks` blocks of 512 threads -- one per CU; launch_peak_fill puts
// random floats there first.
hipError_t launch_read_peak(const void *buf, size_t bytes, float *sink, uint32_t blocks, hipStream_t s);
size_t read_peak_bytes(size_t bytes);
hipError_t launch_peak_fill(void *buf, size_t bytes, hipStream_t s);

// normalize_l2 (distances.rs:350-361) on rows: out = (x / sqrt(f64 sum x^2)) as f32.
hipError_t launch_normalize_l2(const float *in, uint32_t n, uint32_t d, float *out, hipStream_t s);

}  // namespace vt
