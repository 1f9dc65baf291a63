// vt_device.h -- launch interface between the host index (vt_index.cpp) and the
// gfx950 kernels (vt_device.hip).  Internal; the public boundary is
// include/vettore_flat.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vt {

// One candidate of a partial / final top-k list.
//   key = orderable(rank_value(metric, raw)) << 32 | id_rank
// Ascending u64 order of `key` == the reference's (rank.total_cmp, id bytes)
// order (flat.rs:34-40): `id_rank` is order-isomorphic to the bytewise order of
// the row's id among the rows of this index.
struct Entry {
  uint64_t key;
  uint32_t row;
  float raw;
};
static_assert(sizeof(Entry) == 16, "Entry is 16 bytes");

constexpr uint64_t kEmptyKey = ~0ull;
constexpr int kTileRows = 32;   // rows per wave tile in the scan kernel
constexpr int kWavesPerBlock = 4;
constexpr int kMaxFusedK = 256; // largest k one scan pass selects

struct ScanArgs {
  const float *X;          // row-major rows, `stride` floats apart, 16-B aligned
  size_t stride;           // floats between rows (multiple of 8)
  const float *q;          // query padded with zeros to ld floats (device)
  const uint32_t *id_rank; // per row (indexed like rows of X); may be null => row index
  const uint32_t *gather;  // optional: scan rows X[gather[i*gather_stride]] for i < n
  uint32_t gather_stride;  // in uint32 units
  uint32_t n;              // rows to scan
  uint32_t d;              // dimensions used (prefix length)
  int metric;              // VT_* metric code
  int order;               // VT_ORDER_*
  uint32_t k;              // 1..kMaxFusedK
  uint64_t lo_key;         // keep only keys > lo_key when has_lo
  int has_lo;
  uint32_t q_nonzero;      // count of query coordinates != 0 (Jaccard)
  Entry *partial;          // [grid_waves][k]
  int *status;             // device int; set to VT_ERR_OVERFLOW on "metric overflow"
};

// LDS bytes the scan kernel needs for dimension d, or 0 if it does not fit.
size_t scan_lds_bytes(uint32_t d);
// Number of waves (= partial lists) a launch with `blocks` blocks produces.
inline uint32_t scan_waves(uint32_t blocks) { return blocks * kWavesPerBlock; }
hipError_t launch_scan(const ScanArgs &a, uint32_t blocks, hipStream_t s);

// Selects the k smallest keys of in[0..m) (kEmptyKey entries ignored), writes
// them sorted ascending to out[0..*out_count).
hipError_t launch_merge(const Entry *in, uint32_t m, uint32_t k, Entry *out,
                        uint32_t *out_count, hipStream_t s);

struct HammingArgs {
  const uint64_t *bits;    // [n][words]
  const uint64_t *qbits;   // [words] (device)
  const uint32_t *id_rank; // per row or null
  uint32_t n, words, d, k;
  uint64_t lo_key;
  int has_lo;
  Entry *partial;
};
size_t hamming_lds_bytes(uint32_t words);
hipError_t launch_hamming(const HammingArgs &a, uint32_t blocks, hipStream_t s);

// rows[n][stride] (first d columns) -> bits[n][ceil(d/64)], bit j%64 of word
// j/64 set iff v[j] >= 0.0 (distances.rs:413-423).
hipError_t launch_sign_pack(const float *rows, size_t stride, uint32_t n,
                            uint32_t d, uint64_t *bits, hipStream_t s);

// *flag |= 1 if any of the first d columns of any row is non-finite.
hipError_t launch_check_finite(const float *rows, size_t stride, uint32_t n,
                               uint32_t d, int *flag, hipStream_t s);

// dst[n][dst_stride] <- src[n][d], columns d..dst_stride-1 zero-filled.
hipError_t launch_pad_rows(const float *src, uint32_t n, uint32_t d, float *dst,
                           size_t dst_stride, hipStream_t s);

// Exact rerank for Metric::Cosine (search.rs:56-60 -> distances.rs:160-177):
// raw = clamp(f64_dot(q,x) / (sqrt(f64_dot(q,q)) * sqrt(f64_dot(x,x)))) as f32,
// one candidate per lane, sequential f64 sums.  Emits Entry{key,row,raw}.
struct CosineRerankArgs {
  const float *X;
  size_t stride;
  const float *q;          // device, >= d floats
  const uint32_t *id_rank; // per row or null
  const uint32_t *gather;  // candidate rows (required)
  uint32_t gather_stride;
  uint32_t n;              // candidates
  uint32_t d;
  Entry *out;              // [n]
  int *status;
};
hipError_t launch_cosine_rerank(const CosineRerankArgs &a, hipStream_t s);

// normalize_l2 (distances.rs:350-361) on rows: out = (x / sqrt(f64 sum x^2)) as f32.
hipError_t launch_normalize_l2(const float *in, uint32_t n, uint32_t d, float *out,
                               hipStream_t s);

}  // namespace vt
