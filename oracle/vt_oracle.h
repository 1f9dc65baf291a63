/*
 * vt_oracle.h -- CPU ORACLE for the Vettore flat-index hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it, and only as the
 * checker.  The product library (libvettore_hip.so) never links or calls it.
 *
 * It restates, in plain C, the algorithm of the reference Rust crate
 * (/root/reference/native/vettore/src/{distances,flat,search}.rs, v0.3.2).
 * Every function cites the reference lines it follows.
 *
 * Parity status: pinned against every known-answer test the reference holds
 * for this path (tests/golden/).  One thing is NOT
 * pinned by any reference test or on-disk source: the lane order of
 * wide::f32x8::reduce_add (third-party crate `wide` 1.5.0, Cargo.lock:139-147,
 * not vendored).  The reference's own tests check that boundary only to 2e-6
 * relative (distances.rs:570-609).  The oracle therefore implements the three
 * orders a `wide` build can produce and lets the caller pick one; see
 * vto_set_reduce_order().  "summation order: parity unpinned" -- DESIGN.md.
 */
#ifndef VT_ORACLE_H
#define VT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Metric codes: distances.rs:24-38 (Metric::from_code). */
enum {
  VTO_L2 = 0,
  VTO_L2_SQUARED = 1,
  VTO_COSINE = 2,
  VTO_INNER_PRODUCT = 3,
  VTO_NEG_INNER_PRODUCT = 4,
  VTO_MANHATTAN = 5,
  VTO_CHEBYSHEV = 6,
  VTO_HAMMING = 7,
  VTO_JACCARD = 8
};

/* Horizontal-add order of one 8-lane chunk (wide::f32x8::reduce_add).
 *  PAIR: ((l0+l1)+(l2+l3)) + ((l4+l5)+(l6+l7))  non-AVX x86 (SSE shuffles /
 *        hadd) and aarch64 NEON vaddvq: the published precompiled NIFs
 *        (.github/workflows/main.yml sets no RUSTFLAGS).
 *  AVX : ((l0+l4)+(l2+l6)) + ((l1+l5)+(l3+l7))  target-cpu=native builds
 *        (Taskfile.yml:12).
 *  SEQ : (((l0+l1)+l2)+l3) + (((l4+l5)+l6)+l7)  scalar-fallback f32x4.
 */
enum { VTO_ORDER_PAIR = 0, VTO_ORDER_AVX = 1, VTO_ORDER_SEQ = 2, VTO_ORDER_SSE2 = 3 };

/* Error codes; vto_strerror() gives the reference's exact error strings. */
enum {
  VTO_OK = 0,
  VTO_ERR_EMPTY = 1,           /* "vector must not be empty"           flat.rs:138 */
  VTO_ERR_DIMENSION = 2,       /* "dimension mismatch"                 flat.rs:141 */
  VTO_ERR_NON_FINITE = 3,      /* "vector contains a non-finite value" distances.rs:135 */
  VTO_ERR_OVERFLOW = 4,        /* "metric overflow"                    distances.rs:67 */
  VTO_ERR_UNKNOWN_METRIC = 5,  /* "unknown metric"                     distances.rs:36 */
  VTO_ERR_PREFIX = 6,          /* "invalid prefix dimensions"          search.rs:47 */
  VTO_ERR_DIMS_POSITIVE = 7,   /* "dimensions must be positive"        distances.rs:463 */
  VTO_ERR_NOMEM = 8
};

const char *vto_strerror(int code);

void vto_set_reduce_order(int order);
int vto_get_reduce_order(void);

/* distances.rs:42-68 compute(); lengths passed separately so the
 * "dimension mismatch" check is reproduced. */
int vto_compute(int metric, const float *left, size_t nleft, const float *right,
                size_t nright, float *out);
/* distances.rs:101-105 compute_checked(). */
int vto_compute_checked(int metric, const float *left, size_t nleft,
                        const float *right, size_t nright, float *out);
/* distances.rs:113-119 rank_value(). */
float vto_rank_value(int metric, float raw);
/* distances.rs:160-177 cosine() (f64). */
int vto_cosine(const float *left, size_t nleft, const float *right,
               size_t nright, float *out);
/* distances.rs:131-137. */
int vto_validate_finite(const float *v, size_t n);
/* distances.rs:350-361 normalize_l2(). out has n floats. */
int vto_normalize_l2(const float *in, size_t n, float *out);
/* distances.rs:413-423. words has (n+63)/64 entries. */
void vto_compress_sign_bits(const float *v, size_t n, uint64_t *words);
/* distances.rs:426-437 / 440-457. */
int vto_packed_hamming(const uint64_t *left, size_t nleft, const uint64_t *right,
                       size_t nright, size_t dimensions, float *out);
int vto_packed_jaccard(const uint64_t *left, size_t nleft,
                       const uint64_t *right, size_t nright, size_t dimensions,
                       float *out);

/* Result list: Vec<(String, f32)>. */
typedef struct vto_hits vto_hits;
size_t vto_hits_len(const vto_hits *h);
const char *vto_hits_id(const vto_hits *h, size_t i, size_t *len);
float vto_hits_raw(const vto_hits *h, size_t i);
void vto_hits_free(vto_hits *h);

/* flat.rs:13-129 FlatIndex: a hash map of separately allocated rows, like the
 * reference's HashMap<String, Vec<f32>>. */
typedef struct vto_flat vto_flat;
vto_flat *vto_flat_new(int metric);
void vto_flat_free(vto_flat *ix);
size_t vto_flat_len(const vto_flat *ix);
long vto_flat_dimension(const vto_flat *ix); /* -1 = None */
int vto_flat_insert(vto_flat *ix, const char *id, size_t idlen, const float *v,
                    size_t n);
/* Ragged batch: ids concatenated, id_off[count+1]; vals concatenated,
 * val_off[count+1] (in floats). flat.rs:69-85. */
int vto_flat_insert_many(vto_flat *ix, size_t count, const char *ids,
                         const size_t *id_off, const float *vals,
                         const size_t *val_off);
void vto_flat_delete(vto_flat *ix, const char *id, size_t idlen);
int vto_flat_search(const vto_flat *ix, const float *q, size_t nq, size_t limit,
                    vto_hits **out);

/* search.rs:38-73 and 76-92. */
int vto_vector_top_k(size_t count, const char *ids, const size_t *id_off,
                     const float *vals, const size_t *val_off, const float *q,
                     size_t nq, int metric_code, size_t dimensions, size_t limit,
                     vto_hits **out);
int vto_binary_top_k(size_t count, const char *ids, const size_t *id_off,
                     const uint64_t *words, const size_t *word_off,
                     const uint64_t *q, size_t nq, size_t dimensions,
                     size_t limit, vto_hits **out);

/* Convenience for large parity cases: exact flat search over a contiguous
 * row-major matrix with ids given as one concatenated buffer.  Same hits as
 * building a vto_flat from the same rows (ids must be unique) -- it only skips
 * the per-row allocations so that 1e5..1e6-row checks finish in seconds.
 * Used by tests as the checker; the timed cpu_baseline uses vto_flat_search,
 * which keeps the reference's layout. */
int vto_matrix_search(int metric, const float *rows, size_t n, size_t d,
                      const char *ids, const size_t *id_off, const float *q,
                      size_t nq, size_t limit, vto_hits **out);

#ifdef __cplusplus
}
#endif
#endif
