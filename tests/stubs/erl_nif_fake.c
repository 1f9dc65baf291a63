/*
 * erl_nif_fake.c -- a minimal term runtime behind tests/stubs/erl_nif.h, so that
 * integration/c_src/vettore_gpu_nif.c can be EXECUTED in an image without Erlang/OTP
 * (tests/test_nif_exec.py, tests/test_gpu_nif_exec.py).  TEST INFRASTRUCTURE, written from the
 * documented erl_nif API: it implements the ~25 enif_* calls the shim makes over a tagged
 * union (atom / integer / float / binary / list / tuple / resource), keeps resource objects
 * reference-counted with their destructor (what the BEAM's GC does for a ResourceArc), and
 * exports a small fake_* driver API for ctypes: build argument terms, call a NIF by name and
 * arity through the ErlNifFunc table nif_init() returns, walk the result.  It is not OTP, says
 * nothing about schedulers, and is never part of the product.
 */
#include <erl_nif.h>

#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { T_ATOM = 1, T_INT, T_UINT, T_FLOAT, T_BINARY, T_NIL, T_CONS, T_TUPLE, T_RESOURCE, T_BADARG };

typedef struct term {
  int kind;
  struct term *next_in_env; /* arena list */
  union {
    char *atom;
    int64_t i;
    uint64_t u;
    double f;
    struct { unsigned char *data; size_t size; } bin;
    struct { ERL_NIF_TERM head, tail; } cons;
    struct { unsigned n; ERL_NIF_TERM *e; } tuple;
    void *resource; /* user object; header sits in front of it */
  } v;
} term;

struct enif_environment_t {
  term *terms;
  int raised_badarg;
};

struct enif_resource_type_t {
  char name[64];
  ErlNifResourceDtor *dtor;
};

typedef struct {
  ErlNifResourceType *type;
  long refs;
  uint64_t magic;
} res_header;
#define RES_MAGIC 0x7665747446616b65ull

static long g_live_resources = 0, g_dtor_calls = 0;

static term *new_term(ErlNifEnv *env, int kind) {
  term *t = (term *)calloc(1, sizeof *t);
  if (!t) abort();
  t->kind = kind;
  t->next_in_env = env->terms;
  env->terms = t;
  return t;
}
static term *T(ERL_NIF_TERM t) { return (term *)t; }

static res_header *header_of(void *obj) { return (res_header *)obj - 1; }

static void resource_unref(void *obj) {
  res_header *h = header_of(obj);
  if (h->magic != RES_MAGIC) abort();
  if (--h->refs == 0) {
    if (h->type->dtor) h->type->dtor(NULL, obj);
    g_dtor_calls += 1;
    g_live_resources -= 1;
    h->magic = 0;
    free(h);
  }
}

/* ------------------------------------------------------------------ enif_* */
ERL_NIF_TERM enif_make_atom(ErlNifEnv *env, const char *name) {
  term *t = new_term(env, T_ATOM);
  t->v.atom = (char *)malloc(strlen(name) + 1);
  strcpy(t->v.atom, name);
  return (ERL_NIF_TERM)t;
}
ERL_NIF_TERM enif_make_badarg(ErlNifEnv *env) {
  env->raised_badarg = 1;
  return (ERL_NIF_TERM)new_term(env, T_BADARG);
}
ERL_NIF_TERM enif_make_double(ErlNifEnv *env, double d) {
  /* the BEAM has no non-finite floats: enif_make_double of one is a badarg in OTP; the shim never makes one */
  term *t = new_term(env, T_FLOAT);
  t->v.f = d;
  return (ERL_NIF_TERM)t;
}
ERL_NIF_TERM enif_make_uint64(ErlNifEnv *env, ErlNifUInt64 v) {
  term *t = new_term(env, T_UINT);
  t->v.u = v;
  return (ERL_NIF_TERM)t;
}
static ERL_NIF_TERM make_nil(ErlNifEnv *env) { return (ERL_NIF_TERM)new_term(env, T_NIL); }
ERL_NIF_TERM enif_make_list_cell(ErlNifEnv *env, ERL_NIF_TERM head, ERL_NIF_TERM tail) {
  term *t = new_term(env, T_CONS);
  t->v.cons.head = head;
  t->v.cons.tail = tail;
  return (ERL_NIF_TERM)t;
}
ERL_NIF_TERM enif_make_tuple(ErlNifEnv *env, unsigned cnt, ...) {
  term *t = new_term(env, T_TUPLE);
  t->v.tuple.n = cnt;
  t->v.tuple.e = (ERL_NIF_TERM *)calloc(cnt ? cnt : 1, sizeof(ERL_NIF_TERM));
  va_list ap;
  va_start(ap, cnt);
  for (unsigned i = 0; i < cnt; ++i) t->v.tuple.e[i] = va_arg(ap, ERL_NIF_TERM);
  va_end(ap);
  return (ERL_NIF_TERM)t;
}
ERL_NIF_TERM enif_make_list(ErlNifEnv *env, unsigned cnt, ...) {
  ERL_NIF_TERM *e = (ERL_NIF_TERM *)calloc(cnt ? cnt : 1, sizeof(ERL_NIF_TERM));
  va_list ap;
  va_start(ap, cnt);
  for (unsigned i = 0; i < cnt; ++i) e[i] = va_arg(ap, ERL_NIF_TERM);
  va_end(ap);
  ERL_NIF_TERM list = make_nil(env);
  for (unsigned i = cnt; i-- > 0;) list = enif_make_list_cell(env, e[i], list);
  free(e);
  return list;
}
unsigned char *enif_make_new_binary(ErlNifEnv *env, size_t size, ERL_NIF_TERM *termp) {
  term *t = new_term(env, T_BINARY);
  t->v.bin.size = size;
  t->v.bin.data = (unsigned char *)malloc(size ? size : 1);
  *termp = (ERL_NIF_TERM)t;
  return t->v.bin.data;
}

int enif_get_int(ErlNifEnv *env, ERL_NIF_TERM term_, int *ip) {
  (void)env;
  term *t = T(term_);
  if (t->kind == T_INT && t->v.i >= INT32_MIN && t->v.i <= INT32_MAX) { *ip = (int)t->v.i; return 1; }
  if (t->kind == T_UINT && t->v.u <= (uint64_t)INT32_MAX) { *ip = (int)t->v.u; return 1; }
  return 0;
}
int enif_get_long(ErlNifEnv *env, ERL_NIF_TERM term_, long *ip) {
  (void)env;
  term *t = T(term_);
  if (t->kind == T_INT) { *ip = (long)t->v.i; return 1; }
  if (t->kind == T_UINT && t->v.u <= (uint64_t)INT64_MAX) { *ip = (long)t->v.u; return 1; }
  return 0;
}
int enif_get_uint64(ErlNifEnv *env, ERL_NIF_TERM term_, ErlNifUInt64 *ip) {
  (void)env;
  term *t = T(term_);
  if (t->kind == T_UINT) { *ip = t->v.u; return 1; }
  if (t->kind == T_INT && t->v.i >= 0) { *ip = (ErlNifUInt64)t->v.i; return 1; }
  return 0; /* negative integers, floats, bignums beyond 64 bits (not representable here) */
}
int enif_get_double(ErlNifEnv *env, ERL_NIF_TERM term_, double *dp) {
  (void)env;
  term *t = T(term_);
  if (t->kind != T_FLOAT) return 0; /* an integer is NOT a float for enif_get_double */
  *dp = t->v.f;
  return 1;
}
int enif_get_list_length(ErlNifEnv *env, ERL_NIF_TERM term_, unsigned *len) {
  (void)env;
  unsigned n = 0;
  term *t = T(term_);
  while (t->kind == T_CONS) {
    n += 1;
    t = T(t->v.cons.tail);
  }
  if (t->kind != T_NIL) return 0; /* not a list, or an improper one */
  *len = n;
  return 1;
}
int enif_get_list_cell(ErlNifEnv *env, ERL_NIF_TERM list, ERL_NIF_TERM *head, ERL_NIF_TERM *tail) {
  (void)env;
  term *t = T(list);
  if (t->kind != T_CONS) return 0;
  *head = t->v.cons.head;
  *tail = t->v.cons.tail;
  return 1;
}
int enif_get_tuple(ErlNifEnv *env, ERL_NIF_TERM term_, int *arity, const ERL_NIF_TERM **array) {
  (void)env;
  term *t = T(term_);
  if (t->kind != T_TUPLE) return 0;
  *arity = (int)t->v.tuple.n;
  *array = t->v.tuple.e;
  return 1;
}
int enif_inspect_binary(ErlNifEnv *env, ERL_NIF_TERM bin_term, ErlNifBinary *bin) {
  (void)env;
  term *t = T(bin_term);
  if (t->kind != T_BINARY) return 0;
  bin->size = t->v.bin.size;
  bin->data = t->v.bin.data;
  bin->ref_bin = NULL;
  return 1;
}

ErlNifResourceType *enif_open_resource_type(ErlNifEnv *env, const char *module_str, const char *name,
                                            ErlNifResourceDtor *dtor, ErlNifResourceFlags flags,
                                            ErlNifResourceFlags *tried) {
  (void)env; (void)module_str;
  ErlNifResourceType *rt = (ErlNifResourceType *)calloc(1, sizeof *rt);
  snprintf(rt->name, sizeof rt->name, "%s", name);
  rt->dtor = dtor;
  if (tried) *tried = flags;
  return rt;
}
void *enif_alloc_resource(ErlNifResourceType *type, size_t size) {
  res_header *h = (res_header *)calloc(1, sizeof *h + size);
  h->type = type;
  h->refs = 1;
  h->magic = RES_MAGIC;
  g_live_resources += 1;
  return h + 1;
}
void enif_release_resource(void *obj) { resource_unref(obj); }
ERL_NIF_TERM enif_make_resource(ErlNifEnv *env, void *obj) {
  term *t = new_term(env, T_RESOURCE);
  t->v.resource = obj;
  header_of(obj)->refs += 1; /* the term keeps the object alive until its environment goes */
  return (ERL_NIF_TERM)t;
}
int enif_get_resource(ErlNifEnv *env, ERL_NIF_TERM term_, ErlNifResourceType *type, void **objp) {
  (void)env;
  term *t = T(term_);
  if (t->kind != T_RESOURCE || header_of(t->v.resource)->type != type) return 0;
  *objp = t->v.resource;
  return 1;
}

/* ------------------------------------------------------------ driver (ctypes) */
ErlNifEntry *nif_init(void);
static ErlNifEntry *g_entry;

ErlNifEnv *fake_env_new(void) { return (ErlNifEnv *)calloc(1, sizeof(ErlNifEnv)); }

/* frees every term of the environment; resource terms drop their reference (the GC finding the
 * last reference gone is what runs a resource's destructor on the BEAM) */
void fake_env_free(ErlNifEnv *env) {
  term *t = env->terms;
  while (t) {
    term *next = t->next_in_env;
    if (t->kind == T_ATOM) free(t->v.atom);
    if (t->kind == T_BINARY) free(t->v.bin.data);
    if (t->kind == T_TUPLE) free(t->v.tuple.e);
    if (t->kind == T_RESOURCE) resource_unref(t->v.resource);
    free(t);
    t = next;
  }
  free(env);
}

/* module load: nif_init() + the entry's load callback; returns the number of NIFs or -1 */
int fake_load(void) {
  g_entry = nif_init();
  ErlNifEnv *env = fake_env_new();
  void *priv = NULL;
  const int rc = g_entry->load ? g_entry->load(env, &priv, make_nil(env)) : 0;
  fake_env_free(env);
  return rc == 0 ? g_entry->num_of_funcs : -1;
}
const char *fake_module_name(void) { return g_entry ? g_entry->name : ""; }
const char *fake_func_name(int i) { return g_entry->funcs[i].name; }
unsigned fake_func_arity(int i) { return g_entry->funcs[i].arity; }
unsigned fake_func_flags(int i) { return g_entry->funcs[i].flags; }

/* calls NIF `name`/`argc`; 0 = no such function (what :erlang.nif_error would be).  A result of
 * kind T_BADARG stands for the ArgumentError the BEAM raises after enif_make_badarg. */
ERL_NIF_TERM fake_call(ErlNifEnv *env, const char *name, int argc, const ERL_NIF_TERM *argv) {
  for (int i = 0; i < g_entry->num_of_funcs; ++i)
    if (strcmp(g_entry->funcs[i].name, name) == 0 && (int)g_entry->funcs[i].arity == argc) {
      env->raised_badarg = 0;
      ERL_NIF_TERM r = g_entry->funcs[i].fptr(env, argc, argv);
      if (env->raised_badarg) return (ERL_NIF_TERM)new_term(env, T_BADARG); /* an exception wins over any return value */
      return r;
    }
  return 0;
}

ERL_NIF_TERM fake_atom(ErlNifEnv *env, const char *name) { return enif_make_atom(env, name); }
ERL_NIF_TERM fake_int(ErlNifEnv *env, int64_t v) {
  term *t = new_term(env, T_INT);
  t->v.i = v;
  return (ERL_NIF_TERM)t;
}
ERL_NIF_TERM fake_uint(ErlNifEnv *env, uint64_t v) { return enif_make_uint64(env, v); }
ERL_NIF_TERM fake_float(ErlNifEnv *env, double v) { return enif_make_double(env, v); }
ERL_NIF_TERM fake_binary(ErlNifEnv *env, const void *p, size_t n) {
  ERL_NIF_TERM t;
  unsigned char *dst = enif_make_new_binary(env, n, &t);
  if (n) memcpy(dst, p, n);
  return t;
}
ERL_NIF_TERM fake_nil(ErlNifEnv *env) { return make_nil(env); }
ERL_NIF_TERM fake_cons(ErlNifEnv *env, ERL_NIF_TERM head, ERL_NIF_TERM tail) { return enif_make_list_cell(env, head, tail); }
ERL_NIF_TERM fake_list(ErlNifEnv *env, unsigned n, const ERL_NIF_TERM *e) {
  ERL_NIF_TERM list = make_nil(env);
  for (unsigned i = n; i-- > 0;) list = enif_make_list_cell(env, e[i], list);
  return list;
}
/* [float] straight from a double array (long vectors without one ctypes call per element) */
ERL_NIF_TERM fake_float_list(ErlNifEnv *env, unsigned n, const double *v) {
  ERL_NIF_TERM list = make_nil(env);
  for (unsigned i = n; i-- > 0;) list = enif_make_list_cell(env, enif_make_double(env, v[i]), list);
  return list;
}
ERL_NIF_TERM fake_tuple(ErlNifEnv *env, unsigned n, const ERL_NIF_TERM *e) {
  term *t = new_term(env, T_TUPLE);
  t->v.tuple.n = n;
  t->v.tuple.e = (ERL_NIF_TERM *)calloc(n ? n : 1, sizeof(ERL_NIF_TERM));
  for (unsigned i = 0; i < n; ++i) t->v.tuple.e[i] = e[i];
  return (ERL_NIF_TERM)t;
}
/* the same resource object as a term of another environment (a reference held by a process) */
ERL_NIF_TERM fake_copy_resource(ErlNifEnv *env, ERL_NIF_TERM res) {
  if (T(res)->kind != T_RESOURCE) return 0;
  return enif_make_resource(env, T(res)->v.resource);
}

int fake_kind(ERL_NIF_TERM t) { return T(t)->kind; }
const char *fake_atom_name(ERL_NIF_TERM t) { return T(t)->v.atom; }
int64_t fake_int_value(ERL_NIF_TERM t) { return T(t)->v.i; }
uint64_t fake_uint_value(ERL_NIF_TERM t) { return T(t)->v.u; }
double fake_float_value(ERL_NIF_TERM t) { return T(t)->v.f; }
size_t fake_binary_size(ERL_NIF_TERM t) { return T(t)->v.bin.size; }
const unsigned char *fake_binary_data(ERL_NIF_TERM t) { return T(t)->v.bin.data; }
ERL_NIF_TERM fake_head(ERL_NIF_TERM t) { return T(t)->v.cons.head; }
ERL_NIF_TERM fake_tail(ERL_NIF_TERM t) { return T(t)->v.cons.tail; }
unsigned fake_tuple_arity(ERL_NIF_TERM t) { return T(t)->v.tuple.n; }
ERL_NIF_TERM fake_tuple_element(ERL_NIF_TERM t, unsigned i) { return T(t)->v.tuple.e[i]; }
long fake_live_resources(void) { return g_live_resources; }
long fake_dtor_calls(void) { return g_dtor_calls; }
