/*
 * erl_nif.h -- STAND-IN for the Erlang/OTP header of the same name, for one purpose:
 * `cc -fsyntax-only` of integration/c_src/vettore_gpu_nif.c in an image without OTP
 * (tests/test_nif_shim.py), so the shim cannot drift from include/vettore_flat.h
 * unnoticed.  Written from the documented erl_nif API (types and prototypes of the
 * calls the shim makes, nothing else); it is not OTP's header, defines no behaviour and
 * is never linked.  A real build uses the header that ships with OTP ($ERL_INCLUDE).
 */
#ifndef VETTORE_TEST_STUB_ERL_NIF_H
#define VETTORE_TEST_STUB_ERL_NIF_H

#include <stddef.h>
#include <stdint.h>

typedef uintptr_t ERL_NIF_TERM;
typedef uint64_t ErlNifUInt64;
typedef int64_t ErlNifSInt64;
typedef struct enif_environment_t ErlNifEnv;
typedef struct enif_resource_type_t ErlNifResourceType;

typedef struct {
  size_t size;
  unsigned char *data;
  void *ref_bin;
  void *spare[2];
} ErlNifBinary;

typedef void ErlNifResourceDtor(ErlNifEnv *env, void *obj);
typedef enum { ERL_NIF_RT_CREATE = 1, ERL_NIF_RT_TAKEOVER = 2 } ErlNifResourceFlags;

typedef struct {
  const char *name;
  unsigned arity;
  ERL_NIF_TERM (*fptr)(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[]);
  unsigned flags;
} ErlNifFunc;

#define ERL_NIF_DIRTY_JOB_CPU_BOUND 1
#define ERL_NIF_DIRTY_JOB_IO_BOUND 2

typedef struct {
  int major, minor;
  const char *name;
  int num_of_funcs;
  ErlNifFunc *funcs;
  int (*load)(ErlNifEnv *, void **priv_data, ERL_NIF_TERM load_info);
  int (*reload)(ErlNifEnv *, void **priv_data, ERL_NIF_TERM load_info);
  int (*upgrade)(ErlNifEnv *, void **priv_data, void **old_priv_data, ERL_NIF_TERM load_info);
  void (*unload)(ErlNifEnv *, void *priv_data);
} ErlNifEntry;

#define ERL_NIF_INIT(MODULE, FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD)                         \
  ErlNifEntry *nif_init(void);                                                             \
  ErlNifEntry *nif_init(void) {                                                            \
    static ErlNifEntry entry = {2, 17, #MODULE, (int)(sizeof(FUNCS) / sizeof(FUNCS[0])), \
                                FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD};                     \
    return &entry;                                                                         \
  }

/* term construction */
ERL_NIF_TERM enif_make_atom(ErlNifEnv *env, const char *name);
ERL_NIF_TERM enif_make_badarg(ErlNifEnv *env);
ERL_NIF_TERM enif_make_double(ErlNifEnv *env, double d);
ERL_NIF_TERM enif_make_uint64(ErlNifEnv *env, ErlNifUInt64 v);
ERL_NIF_TERM enif_make_tuple(ErlNifEnv *env, unsigned cnt, ...);
ERL_NIF_TERM enif_make_list(ErlNifEnv *env, unsigned cnt, ...);
ERL_NIF_TERM enif_make_list_cell(ErlNifEnv *env, ERL_NIF_TERM head, ERL_NIF_TERM tail);
unsigned char *enif_make_new_binary(ErlNifEnv *env, size_t size, ERL_NIF_TERM *termp);
#define enif_make_tuple2(env, e1, e2) enif_make_tuple(env, 2, e1, e2)

/* term inspection */
int enif_get_int(ErlNifEnv *env, ERL_NIF_TERM term, int *ip);
int enif_get_long(ErlNifEnv *env, ERL_NIF_TERM term, long *ip);
int enif_get_uint64(ErlNifEnv *env, ERL_NIF_TERM term, ErlNifUInt64 *ip);
int enif_get_double(ErlNifEnv *env, ERL_NIF_TERM term, double *dp);
int enif_get_list_length(ErlNifEnv *env, ERL_NIF_TERM term, unsigned *len);
int enif_get_list_cell(ErlNifEnv *env, ERL_NIF_TERM list, ERL_NIF_TERM *head, ERL_NIF_TERM *tail);
int enif_get_tuple(ErlNifEnv *env, ERL_NIF_TERM term, int *arity, const ERL_NIF_TERM **array);
int enif_inspect_binary(ErlNifEnv *env, ERL_NIF_TERM bin_term, ErlNifBinary *bin);

/* resource objects */
ErlNifResourceType *enif_open_resource_type(ErlNifEnv *env, const char *module_str, const char *name,
                                            ErlNifResourceDtor *dtor, ErlNifResourceFlags flags,
                                            ErlNifResourceFlags *tried);
void *enif_alloc_resource(ErlNifResourceType *type, size_t size);
void enif_release_resource(void *obj);
ERL_NIF_TERM enif_make_resource(ErlNifEnv *env, void *obj);
int enif_get_resource(ErlNifEnv *env, ERL_NIF_TERM term, ErlNifResourceType *type, void **objp);

#endif
