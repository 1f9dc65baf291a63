/*
 * vt_oracle.c -- CPU ORACLE (test infrastructure only; see vt_oracle.h).
 *
 * Plain-C restatement of the reference algorithm for the flat-index hot path:
 *   /root/reference/native/vettore/src/distances.rs
 *   /root/reference/native/vettore/src/flat.rs
 *   /root/reference/native/vettore/src/search.rs
 * Build with -ffp-contract=off: rustc never contracts a*b+c into an FMA.
 */
#include "vt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int g_order = VTO_ORDER_SSE2;

void vto_set_reduce_order(int order) {
  if (order >= VTO_ORDER_PAIR && order <= VTO_ORDER_SSE2) g_order = order;
}
int vto_get_reduce_order(void) { return g_order; }

const char *vto_strerror(int code) {
  switch (code) {
    case VTO_OK: return "ok";
    case VTO_ERR_EMPTY: return "vector must not be empty";
    case VTO_ERR_DIMENSION: return "dimension mismatch";
    case VTO_ERR_NON_FINITE: return "vector contains a non-finite value";
    case VTO_ERR_OVERFLOW: return "metric overflow";
    case VTO_ERR_UNKNOWN_METRIC: return "unknown metric";
    case VTO_ERR_PREFIX: return "invalid prefix dimensions";
    case VTO_ERR_DIMS_POSITIVE: return "dimensions must be positive";
    case VTO_ERR_NOMEM: return "out of memory";
    default: return "unknown error";
  }
}

/* ---- wide::f32x8::reduce_add, four possible lane orders (header) --------- */
static inline float reduce_add8(const float l[8]) {
  switch (g_order) {
    case VTO_ORDER_AVX:
      return ((l[0] + l[4]) + (l[2] + l[6])) + ((l[1] + l[5]) + (l[3] + l[7]));
    case VTO_ORDER_SEQ:
      return (((l[0] + l[1]) + l[2]) + l[3]) + (((l[4] + l[5]) + l[6]) + l[7]);
    case VTO_ORDER_SSE2:
      return ((l[0] + l[2]) + (l[1] + l[3])) + ((l[4] + l[6]) + (l[5] + l[7]));
    default:
      return ((l[0] + l[1]) + (l[2] + l[3])) + ((l[4] + l[5]) + (l[6] + l[7]));
  }
}

/* distances.rs:236-270 simd_dot */
static float simd_dot(const float *a, const float *b, size_t n) {
  float acc = 0.0f;
  size_t i = 0;
  while (i + 8 <= n) {
    float p[8];
    for (int j = 0; j < 8; ++j) p[j] = a[i + j] * b[i + j];
    acc += reduce_add8(p);
    i += 8;
  }
  while (i < n) {
    acc += a[i] * b[i];
    i += 1;
  }
  return acc;
}

/* distances.rs:197-233 simd_l2_squared */
static float simd_l2_squared(const float *a, const float *b, size_t n) {
  float acc = 0.0f;
  size_t i = 0;
  while (i + 8 <= n) {
    float p[8];
    for (int j = 0; j < 8; ++j) {
      float diff = a[i + j] - b[i + j];
      p[j] = diff * diff;
    }
    acc += reduce_add8(p);
    i += 8;
  }
  while (i < n) {
    float diff = a[i] - b[i];
    acc += diff * diff;
    i += 1;
  }
  return acc;
}

/* distances.rs:273-308 manhattan */
static float manhattan(const float *a, const float *b, size_t n) {
  float acc = 0.0f;
  size_t i = 0;
  while (i + 8 <= n) {
    float p[8];
    for (int j = 0; j < 8; ++j) p[j] = fabsf(a[i + j] - b[i + j]);
    acc += reduce_add8(p);
    i += 8;
  }
  while (i < n) {
    acc += fabsf(a[i] - b[i]);
    i += 1;
  }
  return acc;
}

/* distances.rs:311-316 chebyshev: fold(0.0, f32::max) */
static float chebyshev(const float *a, const float *b, size_t n) {
  float m = 0.0f;
  for (size_t i = 0; i < n; ++i) m = fmaxf(m, fabsf(a[i] - b[i]));
  return m;
}

/* distances.rs:319-324 hamming over truthiness */
static float hamming(const float *a, const float *b, size_t n) {
  size_t c = 0;
  for (size_t i = 0; i < n; ++i)
    if ((a[i] != 0.0f) != (b[i] != 0.0f)) c++;
  return (float)c;
}

/* distances.rs:327-347 jaccard over truthiness */
static float jaccard(const float *a, const float *b, size_t n) {
  size_t inter = 0, uni = 0;
  for (size_t i = 0; i < n; ++i) {
    int l = a[i] != 0.0f, r = b[i] != 0.0f;
    if (l || r) uni++;
    if (l && r) inter++;
  }
  if (uni == 0) return 0.0f;
  return 1.0f - (float)inter / (float)uni;
}

/* distances.rs:179-194 */
static double f64_dot(const float *a, const float *b, size_t n) {
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += (double)a[i] * (double)b[i];
  return s;
}
static double f64_l2_squared(const float *a, const float *b, size_t n) {
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) {
    double d = (double)a[i] - (double)b[i];
    s += d * d;
  }
  return s;
}

/* distances.rs:140-147 l2 */
static float l2(const float *a, const float *b, size_t n) {
  float sq = simd_l2_squared(a, b, n);
  if (isfinite(sq)) return sqrtf(sq);
  return (float)sqrt(f64_l2_squared(a, b, n));
}

/* distances.rs:92-98 f64_to_f32 */
static int f64_to_f32(double v, float *out) {
  if (isfinite(v) && v >= (double)(-FLT_MAX) && v <= (double)FLT_MAX) {
    *out = (float)v;
    return 1;
  }
  return 0;
}

/* distances.rs:70-90 recover_metric_overflow */
static int recover_metric_overflow(int metric, const float *a, const float *b,
                                   size_t n, float *out) {
  double r;
  switch (metric) {
    case VTO_L2: r = sqrt(f64_l2_squared(a, b, n)); break;
    case VTO_L2_SQUARED: r = f64_l2_squared(a, b, n); break;
    case VTO_COSINE:
    case VTO_INNER_PRODUCT: r = f64_dot(a, b, n); break;
    case VTO_NEG_INNER_PRODUCT: r = -f64_dot(a, b, n); break;
    case VTO_MANHATTAN: {
      double s = 0.0;
      for (size_t i = 0; i < n; ++i) s += fabs((double)a[i] - (double)b[i]);
      r = s;
      break;
    }
    case VTO_CHEBYSHEV: {
      double m = 0.0;
      for (size_t i = 0; i < n; ++i) m = fmax(m, fabs((double)a[i] - (double)b[i]));
      r = m;
      break;
    }
    default: return 0; /* Hamming | Jaccard => None */
  }
  return f64_to_f32(r, out);
}

/* distances.rs:42-68 compute */
int vto_compute(int metric, const float *a, size_t na, const float *b, size_t nb,
                float *out) {
  if (metric < 0 || metric > VTO_JACCARD) return VTO_ERR_UNKNOWN_METRIC;
  if (na != nb) return VTO_ERR_DIMENSION;
  float v;
  switch (metric) {
    case VTO_L2: v = l2(a, b, na); break;
    case VTO_L2_SQUARED: v = simd_l2_squared(a, b, na); break;
    case VTO_COSINE: v = simd_dot(a, b, na); break;
    case VTO_INNER_PRODUCT: v = simd_dot(a, b, na); break;
    case VTO_NEG_INNER_PRODUCT: v = -simd_dot(a, b, na); break;
    case VTO_MANHATTAN: v = manhattan(a, b, na); break;
    case VTO_CHEBYSHEV: v = chebyshev(a, b, na); break;
    case VTO_HAMMING: v = hamming(a, b, na); break;
    default: v = jaccard(a, b, na); break;
  }
  if (isfinite(v)) {
    *out = v;
    return VTO_OK;
  }
  if (recover_metric_overflow(metric, a, b, na, out)) return VTO_OK;
  return VTO_ERR_OVERFLOW;
}

/* distances.rs:131-137 */
int vto_validate_finite(const float *v, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!isfinite(v[i])) return VTO_ERR_NON_FINITE;
  return VTO_OK;
}

/* distances.rs:101-105 */
int vto_compute_checked(int metric, const float *a, size_t na, const float *b,
                        size_t nb, float *out) {
  int rc = vto_validate_finite(a, na);
  if (rc) return rc;
  rc = vto_validate_finite(b, nb);
  if (rc) return rc;
  return vto_compute(metric, a, na, b, nb, out);
}

/* distances.rs:113-119 */
float vto_rank_value(int metric, float raw) {
  switch (metric) {
    case VTO_COSINE: return 1.0f - raw;
    case VTO_INNER_PRODUCT: return -raw;
    default: return raw;
  }
}

/* distances.rs:160-177 */
int vto_cosine(const float *a, size_t na, const float *b, size_t nb, float *out) {
  if (na != nb) return VTO_ERR_DIMENSION;
  double ln = sqrt(f64_dot(a, a, na));
  double rn = sqrt(f64_dot(b, b, na));
  if (ln == 0.0 || rn == 0.0) {
    *out = 0.0f;
    return VTO_OK;
  }
  double sim = f64_dot(a, b, na) / (ln * rn);
  if (!isfinite(sim)) return VTO_ERR_OVERFLOW;
  if (sim < -1.0) sim = -1.0;
  if (sim > 1.0) sim = 1.0;
  *out = (float)sim;
  return VTO_OK;
}

/* distances.rs:350-361 */
int vto_normalize_l2(const float *in, size_t n, float *out) {
  int rc = vto_validate_finite(in, n);
  if (rc) return rc;
  double norm = sqrt(f64_dot(in, in, n));
  if (norm == 0.0) {
    for (size_t i = 0; i < n; ++i) out[i] = 0.0f;
  } else {
    for (size_t i = 0; i < n; ++i) out[i] = (float)((double)in[i] / norm);
  }
  return VTO_OK;
}

/* distances.rs:413-423 */
void vto_compress_sign_bits(const float *v, size_t n, uint64_t *words) {
  size_t nw = (n + 63) / 64;
  for (size_t w = 0; w < nw; ++w) words[w] = 0;
  for (size_t i = 0; i < n; ++i)
    if (v[i] >= 0.0f) words[i / 64] |= (uint64_t)1 << (i % 64);
}

/* distances.rs:459-481 */
static int validate_packed_pair(size_t nleft, size_t nright, size_t dimensions) {
  size_t words = (dimensions + 63) / 64;
  if (dimensions == 0) return VTO_ERR_DIMS_POSITIVE;
  if (nleft != words || nright != words) return VTO_ERR_DIMENSION;
  return VTO_OK;
}
static uint64_t word_mask(size_t index, size_t dimensions) {
  size_t words = (dimensions + 63) / 64;
  size_t rem = dimensions % 64;
  if (index + 1 == words && rem != 0) return ((uint64_t)1 << rem) - 1;
  return UINT64_MAX;
}

/* distances.rs:426-437 */
int vto_packed_hamming(const uint64_t *a, size_t na, const uint64_t *b, size_t nb,
                       size_t dimensions, float *out) {
  int rc = validate_packed_pair(na, nb, dimensions);
  if (rc) return rc;
  uint64_t dist = 0;
  for (size_t i = 0; i < na; ++i)
    dist += (uint64_t)__builtin_popcountll((a[i] ^ b[i]) & word_mask(i, dimensions));
  *out = (float)dist;
  return VTO_OK;
}

/* distances.rs:440-457 */
int vto_packed_jaccard(const uint64_t *a, size_t na, const uint64_t *b, size_t nb,
                       size_t dimensions, float *out) {
  int rc = validate_packed_pair(na, nb, dimensions);
  if (rc) return rc;
  uint64_t inter = 0, uni = 0;
  for (size_t i = 0; i < na; ++i) {
    uint64_t m = word_mask(i, dimensions);
    inter += (uint64_t)__builtin_popcountll((a[i] & b[i]) & m);
    uni += (uint64_t)__builtin_popcountll((a[i] | b[i]) & m);
  }
  *out = uni == 0 ? 0.0f : 1.0f - (float)inter / (float)uni;
  return VTO_OK;
}

/* ---- hits + the bounded max-heap (flat.rs:20-46, search.rs:9-35) ---------- */

typedef struct {
  char *id; /* owned copy when owns_ids, else borrowed */
  size_t idlen;
  float raw;
  float rank;
} hit_t;

struct vto_hits {
  hit_t *v;
  size_t len;
  int owns_ids;
};

size_t vto_hits_len(const vto_hits *h) { return h ? h->len : 0; }
const char *vto_hits_id(const vto_hits *h, size_t i, size_t *len) {
  *len = h->v[i].idlen;
  return h->v[i].id;
}
float vto_hits_raw(const vto_hits *h, size_t i) { return h->v[i].raw; }
void vto_hits_free(vto_hits *h) {
  if (!h) return;
  if (h->owns_ids)
    for (size_t i = 0; i < h->len; ++i) free(h->v[i].id);
  free(h->v);
  free(h);
}

/* f32::total_cmp */
static inline int32_t total_key(float f) {
  int32_t b;
  memcpy(&b, &f, 4);
  b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
  return b;
}
/* String::cmp = bytewise, shorter prefix first */
static inline int id_cmp(const char *a, size_t la, const char *b, size_t lb) {
  size_t m = la < lb ? la : lb;
  int c = m ? memcmp(a, b, m) : 0;
  if (c) return c;
  return la < lb ? -1 : (la > lb ? 1 : 0);
}
/* FlatHit::cmp: rank.total_cmp then id (flat.rs:34-40). raw is ignored. */
static inline int hit_cmp(const hit_t *x, const hit_t *y) {
  int32_t kx = total_key(x->rank), ky = total_key(y->rank);
  if (kx != ky) return kx < ky ? -1 : 1;
  return id_cmp(x->id, x->idlen, y->id, y->idlen);
}

/* std::collections::BinaryHeap, restated so that even the unspecified order of
 * Ord-equal elements (duplicate ids with equal rank) follows the reference. */
typedef struct {
  hit_t *d;
  size_t len, cap;
} heap_t;

static int heap_reserve(heap_t *h, size_t want) {
  if (want <= h->cap) return 1;
  size_t nc = h->cap ? h->cap * 2 : 16;
  if (nc < want) nc = want;
  hit_t *nd = (hit_t *)realloc(h->d, nc * sizeof(hit_t));
  if (!nd) return 0;
  h->d = nd;
  h->cap = nc;
  return 1;
}
static size_t heap_sift_up(heap_t *h, size_t start, size_t pos) {
  hit_t elt = h->d[pos];
  while (pos > start) {
    size_t parent = (pos - 1) / 2;
    if (hit_cmp(&elt, &h->d[parent]) <= 0) break;
    h->d[pos] = h->d[parent];
    pos = parent;
  }
  h->d[pos] = elt;
  return pos;
}
static void heap_sift_down_to_bottom(heap_t *h, size_t pos) {
  size_t end = h->len, start = pos;
  hit_t elt = h->d[pos];
  size_t child = 2 * pos + 1;
  size_t lim = end >= 2 ? end - 2 : 0;
  while (child <= lim && end >= 2) {
    if (hit_cmp(&h->d[child], &h->d[child + 1]) <= 0) child += 1;
    h->d[pos] = h->d[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (child == end - 1) {
    h->d[pos] = h->d[child];
    pos = child;
  }
  h->d[pos] = elt;
  heap_sift_up(h, start, pos);
}
static int heap_push(heap_t *h, hit_t x) {
  if (!heap_reserve(h, h->len + 1)) return 0;
  size_t old = h->len;
  h->d[h->len++] = x;
  heap_sift_up(h, 0, old);
  return 1;
}
static hit_t heap_pop(heap_t *h) {
  hit_t item = h->d[--h->len];
  if (h->len > 0) {
    hit_t t = h->d[0];
    h->d[0] = item;
    item = t;
    heap_sift_down_to_bottom(h, 0);
  }
  return item;
}

/* stable merge sort == slice::sort on Ord */
static void merge_sort(hit_t *v, hit_t *tmp, size_t n) {
  if (n < 2) return;
  if (n <= 8) {
    for (size_t i = 1; i < n; ++i) {
      hit_t x = v[i];
      size_t j = i;
      while (j > 0 && hit_cmp(&v[j - 1], &x) > 0) {
        v[j] = v[j - 1];
        --j;
      }
      v[j] = x;
    }
    return;
  }
  size_t m = n / 2;
  merge_sort(v, tmp, m);
  merge_sort(v + m, tmp, n - m);
  memcpy(tmp, v, m * sizeof(hit_t));
  size_t i = 0, j = m, k = 0;
  while (i < m && j < n) {
    if (hit_cmp(&v[j], &tmp[i]) < 0) v[k++] = v[j++];
    else v[k++] = tmp[i++];
  }
  while (i < m) v[k++] = tmp[i++];
}

/* push_top_k (search.rs:94-105) / the inline form in flat.rs:112-117.
 * `clone_id` reproduces flat.rs:107 (`id: id.clone()` for EVERY row). */
static int push_top_k(heap_t *h, hit_t x, size_t limit, int owns) {
  if (limit == 0) {
    if (owns) free(x.id);
    return 1;
  }
  if (h->len < limit) return heap_push(h, x);
  if (hit_cmp(&x, &h->d[0]) < 0) {
    hit_t worst = heap_pop(h);
    if (owns) free(worst.id);
    return heap_push(h, x);
  }
  if (owns) free(x.id);
  return 1;
}

static int sorted_hits(heap_t *h, int owns, vto_hits **out) {
  vto_hits *r = (vto_hits *)calloc(1, sizeof(*r));
  hit_t *tmp = h->len ? (hit_t *)malloc(h->len * sizeof(hit_t)) : NULL;
  if (!r || (h->len && !tmp)) {
    free(r);
    free(tmp);
    return VTO_ERR_NOMEM;
  }
  merge_sort(h->d, tmp, h->len);
  free(tmp);
  r->v = h->d;
  r->len = h->len;
  r->owns_ids = owns;
  *out = r;
  return VTO_OK;
}

static void heap_abort(heap_t *h, int owns) {
  if (owns)
    for (size_t i = 0; i < h->len; ++i) free(h->d[i].id);
  free(h->d);
}

static char *dup_id(const char *id, size_t len) {
  char *p = (char *)malloc(len ? len : 1);
  if (p && len) memcpy(p, id, len);
  return p;
}

/* ---- FlatIndex (flat.rs:13-129): a hash map of separately allocated rows ---- */

typedef struct {
  char *id;
  size_t idlen;
  float *vec;
  size_t n;
  int state; /* 0 empty, 1 full, 2 tombstone */
} slot_t;

struct vto_flat {
  int metric;
  slot_t *slots;
  size_t nslots, used, live;
  long dimension; /* -1 None */
};

static uint64_t fnv1a(const char *s, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= (unsigned char)s[i];
    h *= 1099511628211ull;
  }
  return h;
}

vto_flat *vto_flat_new(int metric) {
  if (metric < 0 || metric > VTO_JACCARD) return NULL;
  vto_flat *ix = (vto_flat *)calloc(1, sizeof(*ix));
  if (!ix) return NULL;
  ix->metric = metric;
  ix->dimension = -1;
  ix->nslots = 64;
  ix->slots = (slot_t *)calloc(ix->nslots, sizeof(slot_t));
  if (!ix->slots) {
    free(ix);
    return NULL;
  }
  return ix;
}

void vto_flat_free(vto_flat *ix) {
  if (!ix) return;
  for (size_t i = 0; i < ix->nslots; ++i)
    if (ix->slots[i].state == 1) {
      free(ix->slots[i].id);
      free(ix->slots[i].vec);
    }
  free(ix->slots);
  free(ix);
}

size_t vto_flat_len(const vto_flat *ix) { return ix->live; }
long vto_flat_dimension(const vto_flat *ix) { return ix->dimension; }

static slot_t *map_find(const vto_flat *ix, const char *id, size_t idlen) {
  size_t mask = ix->nslots - 1;
  size_t i = (size_t)fnv1a(id, idlen) & mask;
  for (;;) {
    slot_t *s = &ix->slots[i];
    if (s->state == 0) return NULL;
    if (s->state == 1 && s->idlen == idlen && (idlen == 0 || memcmp(s->id, id, idlen) == 0))
      return s;
    i = (i + 1) & mask;
  }
}

static int map_grow(vto_flat *ix) {
  size_t on = ix->nslots;
  slot_t *os = ix->slots;
  size_t nn = on * 2;
  slot_t *ns = (slot_t *)calloc(nn, sizeof(slot_t));
  if (!ns) return 0;
  for (size_t i = 0; i < on; ++i) {
    if (os[i].state != 1) continue;
    size_t j = (size_t)fnv1a(os[i].id, os[i].idlen) & (nn - 1);
    while (ns[j].state) j = (j + 1) & (nn - 1);
    ns[j] = os[i];
  }
  free(os);
  ix->slots = ns;
  ix->nslots = nn;
  ix->used = ix->live;
  return 1;
}

/* HashMap::insert (replace on equal key) */
static int map_insert(vto_flat *ix, const char *id, size_t idlen, const float *v, size_t n) {
  float *vec = (float *)malloc(n ? n * sizeof(float) : 1);
  if (!vec) return 0;
  if (n) memcpy(vec, v, n * sizeof(float));
  slot_t *s = map_find(ix, id, idlen);
  if (s) {
    free(s->vec);
    s->vec = vec;
    s->n = n;
    return 1;
  }
  if ((ix->used + 1) * 10 >= ix->nslots * 7)
    if (!map_grow(ix)) {
      free(vec);
      return 0;
    }
  size_t mask = ix->nslots - 1;
  size_t i = (size_t)fnv1a(id, idlen) & mask;
  while (ix->slots[i].state == 1) i = (i + 1) & mask;
  if (ix->slots[i].state == 0) ix->used++;
  ix->slots[i].id = dup_id(id, idlen);
  ix->slots[i].idlen = idlen;
  ix->slots[i].vec = vec;
  ix->slots[i].n = n;
  ix->slots[i].state = 1;
  ix->live++;
  return 1;
}

/* flat.rs:136-144 validate_vector */
static int validate_vector(const float *v, size_t n, long dimension) {
  if (n == 0) return VTO_ERR_EMPTY;
  if (dimension >= 0 && n != (size_t)dimension) return VTO_ERR_DIMENSION;
  return vto_validate_finite(v, n);
}

/* flat.rs:59-66 */
int vto_flat_insert(vto_flat *ix, const char *id, size_t idlen, const float *v, size_t n) {
  int rc = validate_vector(v, n, ix->dimension);
  if (rc) return rc;
  if (ix->dimension < 0) ix->dimension = (long)n;
  return map_insert(ix, id, idlen, v, n) ? VTO_OK : VTO_ERR_NOMEM;
}

/* flat.rs:69-85 */
int vto_flat_insert_many(vto_flat *ix, size_t count, const char *ids,
                         const size_t *id_off, const float *vals,
                         const size_t *val_off) {
  long expected = ix->dimension;
  if (expected < 0 && count > 0) expected = (long)(val_off[1] - val_off[0]);
  for (size_t i = 0; i < count; ++i) {
    int rc = validate_vector(vals + val_off[i], val_off[i + 1] - val_off[i], expected);
    if (rc) return rc;
  }
  for (size_t i = 0; i < count; ++i)
    if (!map_insert(ix, ids + id_off[i], id_off[i + 1] - id_off[i], vals + val_off[i],
                    val_off[i + 1] - val_off[i]))
      return VTO_ERR_NOMEM;
  if (ix->dimension < 0) ix->dimension = expected;
  return VTO_OK;
}

/* flat.rs:88-93 */
void vto_flat_delete(vto_flat *ix, const char *id, size_t idlen) {
  slot_t *s = map_find(ix, id, idlen);
  if (s) {
    free(s->id);
    free(s->vec);
    s->id = NULL;
    s->vec = NULL;
    s->state = 2;
    ix->live--;
  }
  if (ix->live == 0) ix->dimension = -1;
}

/* flat.rs:96-124 */
int vto_flat_search(const vto_flat *ix, const float *q, size_t nq, size_t limit,
                    vto_hits **out) {
  heap_t h = {0, 0, 0};
  *out = NULL;
  if (limit == 0) return sorted_hits(&h, 1, out);
  int rc = validate_vector(q, nq, ix->dimension);
  if (rc) return rc;
  for (size_t i = 0; i < ix->nslots; ++i) {
    const slot_t *s = &ix->slots[i];
    if (s->state != 1) continue;
    float raw;
    rc = vto_compute(ix->metric, q, nq, s->vec, s->n, &raw);
    if (rc) {
      heap_abort(&h, 1);
      return rc;
    }
    hit_t x;
    x.id = dup_id(s->id, s->idlen); /* flat.rs:107 id.clone() per row */
    x.idlen = s->idlen;
    x.raw = raw;
    x.rank = vto_rank_value(ix->metric, raw);
    if (!x.id || !push_top_k(&h, x, limit, 1)) {
      heap_abort(&h, 1);
      return VTO_ERR_NOMEM;
    }
  }
  return sorted_hits(&h, 1, out);
}

/* search.rs:38-73 */
int vto_vector_top_k(size_t count, const char *ids, const size_t *id_off,
                     const float *vals, const size_t *val_off, const float *q,
                     size_t nq, int metric_code, size_t dimensions, size_t limit,
                     vto_hits **out) {
  *out = NULL;
  if (metric_code < 0 || metric_code > VTO_JACCARD) return VTO_ERR_UNKNOWN_METRIC;
  if (dimensions == 0 || dimensions > nq) return VTO_ERR_PREFIX;
  int rc = vto_validate_finite(q, dimensions);
  if (rc) return rc;
  heap_t h = {0, 0, 0};
  for (size_t i = 0; i < count; ++i) {
    const float *v = vals + val_off[i];
    size_t n = val_off[i + 1] - val_off[i];
    if (dimensions > n) {
      heap_abort(&h, 1);
      return VTO_ERR_DIMENSION;
    }
    rc = vto_validate_finite(v, dimensions);
    float raw = 0.0f;
    if (!rc) {
      if (metric_code == VTO_COSINE) rc = vto_cosine(q, dimensions, v, dimensions, &raw);
      else rc = vto_compute(metric_code, q, dimensions, v, dimensions, &raw);
    }
    if (rc) {
      heap_abort(&h, 1);
      return rc;
    }
    hit_t x;
    x.idlen = id_off[i + 1] - id_off[i];
    x.id = dup_id(ids + id_off[i], x.idlen);
    x.raw = raw;
    x.rank = vto_rank_value(metric_code, raw);
    if (!x.id || !push_top_k(&h, x, limit, 1)) {
      heap_abort(&h, 1);
      return VTO_ERR_NOMEM;
    }
  }
  return sorted_hits(&h, 1, out);
}

/* search.rs:76-92 */
int vto_binary_top_k(size_t count, const char *ids, const size_t *id_off,
                     const uint64_t *words, const size_t *word_off,
                     const uint64_t *q, size_t nq, size_t dimensions,
                     size_t limit, vto_hits **out) {
  *out = NULL;
  float raw;
  int rc = vto_packed_hamming(q, nq, q, nq, dimensions, &raw);
  if (rc) return rc;
  heap_t h = {0, 0, 0};
  for (size_t i = 0; i < count; ++i) {
    rc = vto_packed_hamming(q, nq, words + word_off[i], word_off[i + 1] - word_off[i],
                            dimensions, &raw);
    if (rc) {
      heap_abort(&h, 1);
      return rc;
    }
    hit_t x;
    x.idlen = id_off[i + 1] - id_off[i];
    x.id = dup_id(ids + id_off[i], x.idlen);
    x.raw = raw;
    x.rank = raw;
    if (!x.id || !push_top_k(&h, x, limit, 1)) {
      heap_abort(&h, 1);
      return VTO_ERR_NOMEM;
    }
  }
  return sorted_hits(&h, 1, out);
}

/* Same result as vto_flat_search over the same (unique-id) rows; contiguous
 * layout and borrowed ids only so that big parity cases run in seconds. */
int vto_matrix_search(int metric, const float *rows, size_t n, size_t d,
                      const char *ids, const size_t *id_off, const float *q,
                      size_t nq, size_t limit, vto_hits **out) {
  heap_t h = {0, 0, 0};
  *out = NULL;
  if (metric < 0 || metric > VTO_JACCARD) return VTO_ERR_UNKNOWN_METRIC;
  if (limit == 0) return sorted_hits(&h, 0, out);
  int rc = validate_vector(q, nq, n ? (long)d : -1);
  if (rc) return rc;
  for (size_t i = 0; i < n; ++i) {
    float raw;
    rc = vto_compute(metric, q, nq, rows + i * d, d, &raw);
    if (rc) {
      heap_abort(&h, 0);
      return rc;
    }
    hit_t x;
    x.id = (char *)(ids + id_off[i]);
    x.idlen = id_off[i + 1] - id_off[i];
    x.raw = raw;
    x.rank = vto_rank_value(metric, raw);
    if (!push_top_k(&h, x, limit, 0)) {
      heap_abort(&h, 0);
      return VTO_ERR_NOMEM;
    }
  }
  return sorted_hits(&h, 0, out);
}
