/*
 * vettore_gpu_nif.c -- erl_nif shim between Elixir module Vettore.Gpu.Nifs and
 * libvettore_hip.so (include/vettore_flat.h).  It is to the C ABI what
 * native/vettore/src/nifs.rs is to the Rust crate: term decoding/encoding, the
 * resource object, error tuples with the reference's strings
 * (nifs.rs:107-129, :151-175, :200-309).
 *
 * Build where OTP is installed (this repository's build image has no erl_nif.h):
 *   cc -O2 -fPIC -shared -I"$ERL_INCLUDE" -I../../include vettore_gpu_nif.c \
 *      -L../../vettore_amd/lib -lvettore_hip -o priv/vettore_gpu_nif.so
 *
 * Every NIF is a dirty job (nifs.rs marks all of them schedule = "DirtyCpu"): a
 * search blocks its scheduler thread for the duration of the scan.
 */
#include <erl_nif.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "vettore_flat.h"

static ErlNifResourceType *FLAT;
typedef struct { vt_flat *h; } flat_res;

static void flat_dtor(ErlNifEnv *env, void *obj) {
  (void)env;
  vt_flat_free(((flat_res *)obj)->h); /* ResourceArc drop: HBM, streams, pinned staging */
}

/* ---------------------------------------------------------------- terms */
static ERL_NIF_TERM mk_atom(ErlNifEnv *env, const char *a) { return enif_make_atom(env, a); }

static ERL_NIF_TERM mk_binary(ErlNifEnv *env, const char *p, size_t n) {
  ERL_NIF_TERM t;
  memcpy(enif_make_new_binary(env, n, &t), p, n);
  return t;
}

/* {:error, "dimension mismatch"} etc.; device/unsupported errors carry vt_last_error() */
static ERL_NIF_TERM mk_error(ErlNifEnv *env, int st) {
  const char *msg = (st == VT_ERR_DEVICE || st == VT_ERR_UNSUPPORTED) ? vt_last_error() : vt_strerror(st);
  if (st == VT_ERR_ARGUMENT) return enif_make_badarg(env);
  return enif_make_tuple2(env, mk_atom(env, "error"), mk_binary(env, msg, strlen(msg)));
}

/* Ok(()) as rustler encodes it: {:ok, {}} (vector_algorithms_hardening_test.exs:56) */
static ERL_NIF_TERM mk_ok_unit(ErlNifEnv *env) {
  return enif_make_tuple2(env, mk_atom(env, "ok"), enif_make_tuple(env, 0));
}

/* Vec<(String, f32)> -> [{binary, float}]; frees the hits */
static ERL_NIF_TERM hits_to_list(ErlNifEnv *env, vt_hits *h) {
  ERL_NIF_TERM list = enif_make_list(env, 0);
  for (size_t i = vt_hits_len(h); i-- > 0;) {
    size_t len;
    const char *id = vt_hits_id(h, i, &len);
    ERL_NIF_TERM pair = enif_make_tuple2(env, mk_binary(env, id, len), enif_make_double(env, (double)vt_hits_raw(h, i)));
    list = enif_make_list_cell(env, pair, list);
  }
  vt_hits_free(h);
  return list;
}

static ERL_NIF_TERM ok_hits(ErlNifEnv *env, vt_hits *h) {
  return enif_make_tuple2(env, mk_atom(env, "ok"), hits_to_list(env, h));
}

/* [float] -> malloc'ed f32 array.  Anything but a float, or a double outside the f32 range,
 * is a decode failure = badarg, as with rustler's Vec<f32> (SURVEY 8b conventions; the Elixir
 * layer converts integers with `/ 1` before the call, vettore_distance.ex:659-660). */
static int get_f32_list(ErlNifEnv *env, ERL_NIF_TERM list, float **out, size_t *n) {
  unsigned len;
  if (!enif_get_list_length(env, list, &len)) return 0;
  float *v = (float *)malloc((len ? len : 1) * sizeof(float));
  if (!v) return 0;
  ERL_NIF_TERM head, tail = list;
  for (unsigned i = 0; i < len; ++i) {
    double d;
    if (!enif_get_list_cell(env, tail, &head, &tail) || !enif_get_double(env, head, &d)) { free(v); return 0; }
    if (d > 3.4028234663852886e38 || d < -3.4028234663852886e38) { free(v); return 0; }
    v[i] = (float)d;
  }
  *out = v;
  *n = len;
  return 1;
}

static int get_size(ErlNifEnv *env, ERL_NIF_TERM t, size_t *out) {
  ErlNifUInt64 v;
  if (!enif_get_uint64(env, t, &v)) return 0;
  *out = (size_t)v;
  return 1;
}

static int get_size_list(ErlNifEnv *env, ERL_NIF_TERM list, size_t **out, size_t *n) {
  unsigned len;
  if (!enif_get_list_length(env, list, &len)) return 0;
  size_t *v = (size_t *)malloc((len ? len : 1) * sizeof(size_t));
  if (!v) return 0;
  ERL_NIF_TERM head, tail = list;
  for (unsigned i = 0; i < len; ++i) {
    if (!enif_get_list_cell(env, tail, &head, &tail) || !get_size(env, head, &v[i])) { free(v); return 0; }
  }
  *out = v;
  *n = len;
  return 1;
}

static flat_res *get_flat(ErlNifEnv *env, ERL_NIF_TERM t) {
  flat_res *r;
  return enif_get_resource(env, t, FLAT, (void **)&r) ? r : NULL;
}

/* ------------------------------------------------------- index lifecycle */
/* flat_new(metric_code, [device]) -> reference | {:error, binary}
 * nifs.rs:200-257 (one NIF per metric there).  One device: a plain index; several: ONE resource
 * whose rows are spread over those GPUs (vt_flat_new_sharded) -- still one reference, one process. */
static ERL_NIF_TERM flat_new(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  int code;
  unsigned ndev;
  (void)argc;
  if (!enif_get_int(env, argv[0], &code) || !enif_get_list_length(env, argv[1], &ndev) || ndev == 0 || ndev > 64)
    return enif_make_badarg(env);
  int devs[64];
  ERL_NIF_TERM head, tail = argv[1];
  for (unsigned i = 0; i < ndev; ++i)
    if (!enif_get_list_cell(env, tail, &head, &tail) || !enif_get_int(env, head, &devs[i]) || devs[i] < 0)
      return enif_make_badarg(env);
  vt_flat *h;
  int st = vt_flat_new_sharded(code, devs, ndev, &h);
  if (st != VT_OK) return mk_error(env, st);
  flat_res *r = (flat_res *)enif_alloc_resource(FLAT, sizeof *r);
  r->h = h;
  ERL_NIF_TERM t = enif_make_resource(env, r);
  enif_release_resource(r);
  return t;
}

/* flat_insert(ref, id, [float]) -> {:ok, {}} | {:error, binary}        nifs.rs:259-271 */
static ERL_NIF_TERM flat_insert(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  flat_res *r = get_flat(env, argv[0]);
  ErlNifBinary id;
  float *v;
  size_t n;
  (void)argc;
  if (!r || !enif_inspect_binary(env, argv[1], &id) || !get_f32_list(env, argv[2], &v, &n)) return enif_make_badarg(env);
  int st = vt_flat_insert(r->h, (const char *)id.data, id.size, v, n);
  free(v);
  return st == VT_OK ? mk_ok_unit(env) : mk_error(env, st);
}

/* flat_insert_many(ref, [{id, [float]}]) -> {:ok, {}} | {:error, binary}   nifs.rs:273-284
 * (atomic: libvettore_hip validates every row before storing any, flat.rs:69-85) */
static ERL_NIF_TERM flat_insert_many(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  flat_res *r = get_flat(env, argv[0]);
  unsigned count;
  (void)argc;
  if (!r || !enif_get_list_length(env, argv[1], &count)) return enif_make_badarg(env);
  size_t *id_off = (size_t *)calloc(count + 1, sizeof(size_t));
  size_t *val_off = (size_t *)calloc(count + 1, sizeof(size_t));
  size_t ids_cap = 64, vals_cap = 64;
  char *ids = (char *)malloc(ids_cap);
  float *vals = (float *)malloc(vals_cap * sizeof(float));
  int ok = id_off && val_off && ids && vals;
  ERL_NIF_TERM head, tail = argv[1];
  for (unsigned i = 0; ok && i < count; ++i) {
    const ERL_NIF_TERM *pair;
    int arity;
    ErlNifBinary id;
    float *v = NULL;
    size_t n = 0;
    ok = enif_get_list_cell(env, tail, &head, &tail) && enif_get_tuple(env, head, &arity, &pair) && arity == 2 &&
         enif_inspect_binary(env, pair[0], &id) && get_f32_list(env, pair[1], &v, &n);
    if (!ok) break;
    if (id_off[i] + id.size > ids_cap) {
      ids_cap = (id_off[i] + id.size) * 2 + 64;
      ids = (char *)realloc(ids, ids_cap);
    }
    if (val_off[i] + n > vals_cap) {
      vals_cap = (val_off[i] + n) * 2 + 64;
      vals = (float *)realloc(vals, vals_cap * sizeof(float));
    }
    ok = ids && vals;
    if (ok) {
      memcpy(ids + id_off[i], id.data, id.size);
      memcpy(vals + val_off[i], v, n * sizeof(float));
      id_off[i + 1] = id_off[i] + id.size;
      val_off[i + 1] = val_off[i] + n;
    }
    free(v);
  }
  ERL_NIF_TERM res;
  if (!ok) {
    res = enif_make_badarg(env);
  } else {
    int st = vt_flat_insert_many(r->h, count, ids, id_off, vals, val_off);
    res = st == VT_OK ? mk_ok_unit(env) : mk_error(env, st);
  }
  free(ids); free(vals); free(id_off); free(val_off);
  return res;
}

/* flat_load_binary(ref, [id], rows :: binary, d) -> {:ok, {}} | {:error, binary}
 * Bulk ingest without list cells: `rows` is count * d native-endian f32 (e.g. built with
 * <<x::float-32-native>> or Nx.to_binary/1).  Same semantics as insert_many.   */
static ERL_NIF_TERM flat_load_binary(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  flat_res *r = get_flat(env, argv[0]);
  unsigned count;
  ErlNifBinary rows;
  size_t d;
  (void)argc;
  if (!r || !enif_get_list_length(env, argv[1], &count) || !enif_inspect_binary(env, argv[2], &rows) ||
      !get_size(env, argv[3], &d) || rows.size != (size_t)count * d * sizeof(float))
    return enif_make_badarg(env);
  size_t *id_off = (size_t *)calloc(count + 1, sizeof(size_t));
  size_t ids_cap = 64;
  char *ids = (char *)malloc(ids_cap);
  int ok = id_off != NULL && ids != NULL;
  ERL_NIF_TERM head, tail = argv[1];
  for (unsigned i = 0; ok && i < count; ++i) {
    ErlNifBinary id;
    ok = enif_get_list_cell(env, tail, &head, &tail) && enif_inspect_binary(env, head, &id);
    if (!ok) break;
    if (id_off[i] + id.size > ids_cap) {
      ids_cap = (id_off[i] + id.size) * 2 + 64;
      ids = (char *)realloc(ids, ids_cap);
      ok = ids != NULL;
      if (!ok) break;
    }
    memcpy(ids + id_off[i], id.data, id.size);
    id_off[i + 1] = id_off[i] + id.size;
  }
  ERL_NIF_TERM res;
  if (!ok) {
    res = enif_make_badarg(env);
  } else {
    int st = vt_flat_load_matrix(r->h, count, d, ids, id_off, (const float *)rows.data);
    res = st == VT_OK ? mk_ok_unit(env) : mk_error(env, st);
  }
  free(ids); free(id_off);
  return res;
}

/* flat_delete(ref, id) -> {:ok, {}}                                      nifs.rs:286-295 */
static ERL_NIF_TERM flat_delete(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  flat_res *r = get_flat(env, argv[0]);
  ErlNifBinary id;
  (void)argc;
  if (!r || !enif_inspect_binary(env, argv[1], &id)) return enif_make_badarg(env);
  int st = vt_flat_delete(r->h, (const char *)id.data, id.size);
  return st == VT_OK ? mk_ok_unit(env) : mk_error(env, st);
}

/* ----------------------------------------------------------------- search */
/* flat_search(ref, [float], limit) -> {:ok, [{id, raw}]} | {:error, binary}   nifs.rs:297-309 */
static ERL_NIF_TERM flat_search(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  flat_res *r = get_flat(env, argv[0]);
  float *q;
  size_t n, limit;
  (void)argc;
  if (!r || !get_f32_list(env, argv[1], &q, &n) || !get_size(env, argv[2], &limit)) return enif_make_badarg(env);
  vt_hits *h;
  int st = vt_flat_search(r->h, q, n, limit, &h);
  free(q);
  return st == VT_OK ? ok_hits(env, h) : mk_error(env, st);
}

/* flat_search_batch(ref, [[float]], limit) -> {:ok, [[{id, raw}]]}: B x flat_search in one pass
 * over the corpus (FP32 matrix cores + exact rescoring); all queries must have the same length */
static ERL_NIF_TERM flat_search_batch(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  flat_res *r = get_flat(env, argv[0]);
  unsigned nq;
  size_t limit, d = 0;
  (void)argc;
  if (!r || !enif_get_list_length(env, argv[1], &nq) || !get_size(env, argv[2], &limit)) return enif_make_badarg(env);
  if (nq == 0) return enif_make_tuple2(env, mk_atom(env, "ok"), enif_make_list(env, 0));
  float *qs = NULL;
  ERL_NIF_TERM head, tail = argv[1];
  for (unsigned i = 0; i < nq; ++i) {
    float *q;
    size_t n;
    if (!enif_get_list_cell(env, tail, &head, &tail) || !get_f32_list(env, head, &q, &n)) { free(qs); return enif_make_badarg(env); }
    if (i == 0) {
      d = n;
      qs = (float *)malloc((size_t)nq * (d ? d : 1) * sizeof(float));
    }
    if (!qs || n != d) { free(q); free(qs); return n != d ? mk_error(env, VT_ERR_DIMENSION) : enif_make_badarg(env); }
    memcpy(qs + (size_t)i * d, q, d * sizeof(float));
    free(q);
  }
  vt_hits **out = (vt_hits **)calloc(nq, sizeof(vt_hits *));
  int st = out ? vt_flat_search_batch(r->h, qs, nq, d, limit, out) : VT_ERR_NOMEM;
  free(qs);
  if (st != VT_OK) { free(out); return mk_error(env, st); }
  ERL_NIF_TERM list = enif_make_list(env, 0);
  for (unsigned i = nq; i-- > 0;) list = enif_make_list_cell(env, hits_to_list(env, out[i]), list);
  free(out);
  return enif_make_tuple2(env, mk_atom(env, "ok"), list);
}

/* [[float]] -> one malloc'ed [nq][d] f32 matrix; *st = VT_ERR_DIMENSION when the rows differ in length */
static int get_f32_matrix(ErlNifEnv *env, ERL_NIF_TERM list, unsigned nq, float **out, size_t *d, int *st) {
  float *qs = NULL;
  ERL_NIF_TERM head, tail = list;
  *st = VT_OK;
  *d = 0;
  for (unsigned i = 0; i < nq; ++i) {
    float *q;
    size_t n;
    if (!enif_get_list_cell(env, tail, &head, &tail) || !get_f32_list(env, head, &q, &n)) { free(qs); return 0; }
    if (i == 0) {
      *d = n;
      qs = (float *)malloc((size_t)nq * (n ? n : 1) * sizeof(float));
    }
    if (!qs || n != *d) {
      free(q);
      free(qs);
      if (n != *d) { *st = VT_ERR_DIMENSION; return 1; }
      return 0;
    }
    memcpy(qs + (size_t)i * n, q, n * sizeof(float));
    free(q);
  }
  *out = qs;
  return 1;
}

/* [vt_hits *] -> [[{id, raw}]] (frees them) */
static ERL_NIF_TERM hit_lists(ErlNifEnv *env, vt_hits **out, unsigned nq) {
  ERL_NIF_TERM list = enif_make_list(env, 0);
  for (unsigned i = nq; i-- > 0;) list = enif_make_list_cell(env, hits_to_list(env, out[i]), list);
  return enif_make_tuple2(env, mk_atom(env, "ok"), list);
}

/* flat_quantized_search_batch(ref, [[float]], candidates, limit) -> {:ok, [[{id, raw}]]}: B x
 * quantized_search, groups of up to eight sharing one sweep of the sign bits */
static ERL_NIF_TERM flat_quantized_search_batch(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  flat_res *r = get_flat(env, argv[0]);
  unsigned nq;
  size_t candidates, limit, d;
  float *qs;
  int st;
  (void)argc;
  if (!r || !enif_get_list_length(env, argv[1], &nq) || !get_size(env, argv[2], &candidates) || !get_size(env, argv[3], &limit))
    return enif_make_badarg(env);
  if (nq == 0) return enif_make_tuple2(env, mk_atom(env, "ok"), enif_make_list(env, 0));
  if (!get_f32_matrix(env, argv[1], nq, &qs, &d, &st)) return enif_make_badarg(env);
  if (st != VT_OK) return mk_error(env, st);
  vt_hits **out = (vt_hits **)calloc(nq, sizeof(vt_hits *));
  st = out ? vt_flat_quantized_search_batch(r->h, qs, nq, d, candidates, limit, out) : VT_ERR_NOMEM;
  free(qs);
  if (st != VT_OK) { free(out); return mk_error(env, st); }
  ERL_NIF_TERM res = hit_lists(env, out, nq);
  free(out);
  return res;
}

/* flat_quantized_search(ref, [float], candidates, limit)   collection.ex:276-295 in one call */
static ERL_NIF_TERM flat_quantized_search(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  flat_res *r = get_flat(env, argv[0]);
  float *q;
  size_t n, candidates, limit;
  (void)argc;
  if (!r || !get_f32_list(env, argv[1], &q, &n) || !get_size(env, argv[2], &candidates) || !get_size(env, argv[3], &limit))
    return enif_make_badarg(env);
  vt_hits *h;
  int st = vt_flat_quantized_search(r->h, q, n, candidates, limit, &h);
  free(q);
  return st == VT_OK ? ok_hits(env, h) : mk_error(env, st);
}

/* flat_funnel_search(ref, [float], [stage], candidates, limit)   collection.ex:245-260, :674-691 */
static ERL_NIF_TERM flat_funnel_search(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  flat_res *r = get_flat(env, argv[0]);
  float *q;
  size_t n, nst, candidates, limit, *stages;
  (void)argc;
  if (!r || !get_f32_list(env, argv[1], &q, &n)) return enif_make_badarg(env);
  if (!get_size_list(env, argv[2], &stages, &nst) || !get_size(env, argv[3], &candidates) || !get_size(env, argv[4], &limit)) {
    free(q);
    return enif_make_badarg(env);
  }
  vt_hits *h;
  int st = vt_flat_funnel_search(r->h, q, n, stages, nst, candidates, limit, &h);
  free(q); free(stages);
  return st == VT_OK ? ok_hits(env, h) : mk_error(env, st);
}

/* flat_funnel_search_batch(ref, [[float]], [stage], candidates, limit) -> {:ok, [[{id, raw}]]}: B x
 * funnel_search with one set of stages; on a cosine collection groups of up to eight share the
 * stage-1 sweep of the prefixes */
static ERL_NIF_TERM flat_funnel_search_batch(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  flat_res *r = get_flat(env, argv[0]);
  unsigned nq;
  size_t candidates, limit, d, nst, *stages;
  float *qs;
  int st;
  (void)argc;
  if (!r || !enif_get_list_length(env, argv[1], &nq) || !get_size(env, argv[3], &candidates) || !get_size(env, argv[4], &limit))
    return enif_make_badarg(env);
  if (nq == 0) return enif_make_tuple2(env, mk_atom(env, "ok"), enif_make_list(env, 0));
  if (!get_size_list(env, argv[2], &stages, &nst)) return enif_make_badarg(env);
  if (!get_f32_matrix(env, argv[1], nq, &qs, &d, &st)) { free(stages); return enif_make_badarg(env); }
  if (st != VT_OK) { free(stages); return mk_error(env, st); }
  vt_hits **out = (vt_hits **)calloc(nq, sizeof(vt_hits *));
  st = out ? vt_flat_funnel_search_batch(r->h, qs, nq, d, stages, nst, candidates, limit, out) : VT_ERR_NOMEM;
  free(qs); free(stages);
  if (st != VT_OK) { free(out); return mk_error(env, st); }
  ERL_NIF_TERM res = hit_lists(env, out, nq);
  free(out);
  return res;
}

/* flat_hybrid_search(ref, [float], [{kind, candidates, [stage]}], limit)   collection.ex:325-345, :515-592
 * kind: 0 funnel, 1 quantized, 2 search; rerank: :exact */
static ERL_NIF_TERM flat_hybrid_search(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  flat_res *r = get_flat(env, argv[0]);
  float *q;
  size_t n, limit;
  unsigned ngen;
  (void)argc;
  if (!r || !enif_get_list_length(env, argv[2], &ngen) || !get_size(env, argv[3], &limit) || !get_f32_list(env, argv[1], &q, &n))
    return enif_make_badarg(env);
  int *kinds = (int *)calloc(ngen + 1, sizeof(int));
  size_t *cands = (size_t *)calloc(ngen + 1, sizeof(size_t));
  size_t *off = (size_t *)calloc(ngen + 2, sizeof(size_t));
  size_t *stages = NULL, stages_cap = 0;
  int ok = kinds && cands && off;
  ERL_NIF_TERM head, tail = argv[2];
  for (unsigned i = 0; ok && i < ngen; ++i) {
    const ERL_NIF_TERM *t;
    int arity;
    size_t *st = NULL, nst = 0;
    ok = enif_get_list_cell(env, tail, &head, &tail) && enif_get_tuple(env, head, &arity, &t) && arity == 3 &&
         enif_get_int(env, t[0], &kinds[i]) && get_size(env, t[1], &cands[i]) && get_size_list(env, t[2], &st, &nst);
    if (!ok) break;
    if (off[i] + nst > stages_cap) {
      stages_cap = (off[i] + nst) * 2 + 8;
      stages = (size_t *)realloc(stages, stages_cap * sizeof(size_t));
    }
    if (stages) memcpy(stages + off[i], st, nst * sizeof(size_t));
    off[i + 1] = off[i] + nst;
    free(st);
    ok = stages != NULL || off[i + 1] == 0;
  }
  ERL_NIF_TERM res;
  if (!ok) {
    res = enif_make_badarg(env);
  } else {
    size_t dummy = 0;
    vt_hits *h;
    int st = vt_flat_hybrid_search(r->h, q, n, kinds, cands, off, stages ? stages : &dummy, ngen, limit, &h);
    res = st == VT_OK ? ok_hits(env, h) : mk_error(env, st);
  }
  free(q); free(kinds); free(cands); free(off); free(stages);
  return res;
}

/* ------------------------------------------------------ stateless helpers */
/* normalize_l2([float]) -> {:ok, [float]} | {:error, binary}             nifs.rs:107-111 */
static ERL_NIF_TERM normalize_l2(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  float *v;
  size_t n;
  (void)argc;
  if (!get_f32_list(env, argv[0], &v, &n)) return enif_make_badarg(env);
  float *out = (float *)malloc((n ? n : 1) * sizeof(float));
  int st = out ? vt_normalize_l2(0, 1, n, v, out) : VT_ERR_NOMEM;
  ERL_NIF_TERM res;
  if (st != VT_OK) {
    res = mk_error(env, st);
  } else {
    ERL_NIF_TERM list = enif_make_list(env, 0);
    for (size_t i = n; i-- > 0;) list = enif_make_list_cell(env, enif_make_double(env, (double)out[i]), list);
    res = enif_make_tuple2(env, mk_atom(env, "ok"), list);
  }
  free(v); free(out);
  return res;
}

/* compress_sign_bits([float]) -> [u64]   (bare list, as the reference returns it) nifs.rs:113-129 */
static ERL_NIF_TERM compress_sign_bits(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  float *v;
  size_t n;
  (void)argc;
  if (!get_f32_list(env, argv[0], &v, &n)) return enif_make_badarg(env);
  size_t words = (n + 63) / 64;
  uint64_t *out = (uint64_t *)calloc(words ? words : 1, sizeof(uint64_t));
  int st = out ? vt_compress_sign_bits(0, 1, n, v, out) : VT_ERR_NOMEM;
  ERL_NIF_TERM res;
  if (st != VT_OK) {
    res = mk_error(env, st);
  } else {
    res = enif_make_list(env, 0);
    for (size_t i = words; i-- > 0;) res = enif_make_list_cell(env, enif_make_uint64(env, (ErlNifUInt64)out[i]), res);
  }
  free(v); free(out);
  return res;
}

/* [{id, [element]}] -> ids + id_off + values + value_off, as the C ABI's ragged batches expect.
 * `u64` selects [u64] rows (binary_top_k) instead of [float] rows (vector_top_k). */
typedef struct { char *ids; size_t *id_off; void *vals; size_t *val_off; unsigned count; } ragged;

static void ragged_free(ragged *g) { free(g->ids); free(g->id_off); free(g->vals); free(g->val_off); }

static int get_u64_list(ErlNifEnv *env, ERL_NIF_TERM list, uint64_t **out, size_t *n) {
  unsigned len;
  if (!enif_get_list_length(env, list, &len)) return 0;
  uint64_t *v = (uint64_t *)malloc((len ? len : 1) * sizeof(uint64_t));
  if (!v) return 0;
  ERL_NIF_TERM head, tail = list;
  for (unsigned i = 0; i < len; ++i) {
    ErlNifUInt64 w;
    if (!enif_get_list_cell(env, tail, &head, &tail) || !enif_get_uint64(env, head, &w)) { free(v); return 0; }
    v[i] = (uint64_t)w;
  }
  *out = v;
  *n = len;
  return 1;
}

static int get_ragged(ErlNifEnv *env, ERL_NIF_TERM list, int u64, ragged *g) {
  memset(g, 0, sizeof *g);
  if (!enif_get_list_length(env, list, &g->count)) return 0;
  g->id_off = (size_t *)calloc(g->count + 1, sizeof(size_t));
  g->val_off = (size_t *)calloc(g->count + 1, sizeof(size_t));
  if (!g->id_off || !g->val_off) return 0;
  const size_t esz = u64 ? sizeof(uint64_t) : sizeof(float);
  size_t ids_cap = 64, vals_cap = 64;
  g->ids = (char *)malloc(ids_cap);
  g->vals = malloc(vals_cap * esz);
  if (!g->ids || !g->vals) return 0;
  ERL_NIF_TERM head, tail = list;
  for (unsigned i = 0; i < g->count; ++i) {
    const ERL_NIF_TERM *pair;
    int arity;
    ErlNifBinary id;
    void *v = NULL;
    size_t n = 0;
    if (!enif_get_list_cell(env, tail, &head, &tail) || !enif_get_tuple(env, head, &arity, &pair) || arity != 2 ||
        !enif_inspect_binary(env, pair[0], &id))
      return 0;
    if (!(u64 ? get_u64_list(env, pair[1], (uint64_t **)&v, &n) : get_f32_list(env, pair[1], (float **)&v, &n))) return 0;
    if (g->id_off[i] + id.size > ids_cap) {
      ids_cap = (g->id_off[i] + id.size) * 2 + 64;
      g->ids = (char *)realloc(g->ids, ids_cap);
    }
    if (g->val_off[i] + n > vals_cap) {
      vals_cap = (g->val_off[i] + n) * 2 + 64;
      g->vals = realloc(g->vals, vals_cap * esz);
    }
    if (!g->ids || !g->vals) { free(v); return 0; }
    memcpy(g->ids + g->id_off[i], id.data, id.size);
    memcpy((char *)g->vals + g->val_off[i] * esz, v, n * esz);
    g->id_off[i + 1] = g->id_off[i] + id.size;
    g->val_off[i + 1] = g->val_off[i] + n;
    free(v);
  }
  return 1;
}

/* vector_top_k([{id, [float]}], [float], metric_code, dims, limit)        nifs.rs:151-162 */
static ERL_NIF_TERM vector_top_k(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  ragged g;
  float *q = NULL;
  size_t nq, dims, limit;
  int code;
  (void)argc;
  if (!get_ragged(env, argv[0], 0, &g) || !get_f32_list(env, argv[1], &q, &nq) || !enif_get_int(env, argv[2], &code) ||
      !get_size(env, argv[3], &dims) || !get_size(env, argv[4], &limit)) {
    ragged_free(&g);
    free(q);
    return enif_make_badarg(env);
  }
  vt_hits *h;
  int st = vt_vector_top_k(0, g.count, g.ids, g.id_off, (const float *)g.vals, g.val_off, q, nq, code, dims, limit, &h);
  ragged_free(&g);
  free(q);
  return st == VT_OK ? ok_hits(env, h) : mk_error(env, st);
}

/* binary_top_k([{id, [u64]}], [u64], dims, limit)                         nifs.rs:164-175 */
static ERL_NIF_TERM binary_top_k(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]) {
  ragged g;
  uint64_t *q = NULL;
  size_t nq, dims, limit;
  (void)argc;
  if (!get_ragged(env, argv[0], 1, &g) || !get_u64_list(env, argv[1], &q, &nq) || !get_size(env, argv[2], &dims) ||
      !get_size(env, argv[3], &limit)) {
    ragged_free(&g);
    free(q);
    return enif_make_badarg(env);
  }
  vt_hits *h;
  int st = vt_binary_top_k(0, g.count, g.ids, g.id_off, (const uint64_t *)g.vals, g.val_off, q, nq, dims, limit, &h);
  ragged_free(&g);
  free(q);
  return st == VT_OK ? ok_hits(env, h) : mk_error(env, st);
}

static int load(ErlNifEnv *env, void **priv, ERL_NIF_TERM info) {
  (void)priv; (void)info;
  /* a libvettore_hip.so built from another header would be handed structs of the wrong size */
  if (vt_abi_version() != VT_ABI_VERSION) return 1;
  FLAT = enif_open_resource_type(env, NULL, "vettore_gpu_flat", flat_dtor, ERL_NIF_RT_CREATE, NULL);
  return FLAT ? 0 : 1;
}

static ErlNifFunc funcs[] = {
  {"flat_new", 2, flat_new, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flat_insert", 3, flat_insert, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flat_insert_many", 2, flat_insert_many, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flat_load_binary", 4, flat_load_binary, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flat_delete", 2, flat_delete, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flat_search", 3, flat_search, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flat_search_batch", 3, flat_search_batch, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flat_quantized_search", 4, flat_quantized_search, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flat_quantized_search_batch", 4, flat_quantized_search_batch, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flat_funnel_search_batch", 5, flat_funnel_search_batch, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flat_funnel_search", 5, flat_funnel_search, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flat_hybrid_search", 4, flat_hybrid_search, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"normalize_l2", 1, normalize_l2, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"compress_sign_bits", 1, compress_sign_bits, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"vector_top_k", 5, vector_top_k, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"binary_top_k", 4, binary_top_k, ERL_NIF_DIRTY_JOB_IO_BOUND},
};

ERL_NIF_INIT(Elixir.Vettore.Gpu.Nifs, funcs, load, NULL, NULL, NULL)
