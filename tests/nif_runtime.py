"""Runs integration/c_src/vettore_gpu_nif.c without Erlang/OTP: builds it against the fake term
runtime (tests/stubs/erl_nif_fake.c), links libvettore_hip.so, and gives the tests an
Elixir-shaped view of it -- `call("flat_search", ref, [1.0, 0.0], 2)` takes and returns Python
values that stand for terms:

    bytes <-> binary      float <-> float      int <-> integer      list <-> list
    tuple <-> tuple       Atom("ok") <-> :ok   Resource <-> reference (a NIF resource)

enif_make_badarg surfaces as ArgumentError, as it does on the BEAM.  Test infrastructure only.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "integration", "c_src", "vettore_gpu_nif.c")
FAKE = os.path.join(ROOT, "tests", "stubs", "erl_nif_fake.c")
STUBS = os.path.join(ROOT, "tests", "stubs")
BUILD = os.path.join(ROOT, "tests", "_build")
LIBDIR = os.path.join(ROOT, "vettore_amd", "lib")
OUT = os.path.join(BUILD, "vettore_gpu_nif_fake.so")

T_ATOM, T_INT, T_UINT, T_FLOAT, T_BINARY, T_NIL, T_CONS, T_TUPLE, T_RESOURCE, T_BADARG = range(1, 11)


class ArgumentError(Exception):
    """what the BEAM raises after enif_make_badarg"""


class Atom(str):
    def __repr__(self):
        return ":" + str(self)


OK, ERROR = Atom("ok"), Atom("error")


def build():
    srcs = [SHIM, FAKE, os.path.join(STUBS, "erl_nif.h"), os.path.join(ROOT, "include", "vettore_flat.h")]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        return OUT
    os.makedirs(BUILD, exist_ok=True)
    cmd = ["cc", "-std=c11", "-O1", "-g", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", "-I" + STUBS,
           "-I" + os.path.join(ROOT, "include"), SHIM, FAKE, "-L" + LIBDIR, "-lvettore_hip",
           "-Wl,-rpath," + LIBDIR, "-o", OUT]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return OUT


class Resource:
    """A reference to a NIF resource held by "a process": its own environment keeps the object
    alive; release() drops that reference (the GC's part)."""

    def __init__(self, rt, term):
        self.rt = rt
        self.env = rt.L.fake_env_new()
        self.term = rt.L.fake_copy_resource(self.env, term)
        assert self.term

    def release(self):
        if self.env:
            self.rt.L.fake_env_free(self.env)
            self.env = None

    def __del__(self):
        self.release()


class Runtime:
    def __init__(self):
        L = C.CDLL(build())
        vp, term = C.c_void_p, C.c_void_p
        L.fake_env_new.restype = vp
        L.fake_env_free.argtypes = [vp]
        L.fake_load.restype = C.c_int
        L.fake_module_name.restype = C.c_char_p
        L.fake_func_name.restype = C.c_char_p
        L.fake_func_name.argtypes = [C.c_int]
        L.fake_func_arity.argtypes = [C.c_int]
        L.fake_func_flags.argtypes = [C.c_int]
        L.fake_call.restype = term
        L.fake_call.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(term)]
        for name, args in (("fake_atom", [vp, C.c_char_p]), ("fake_int", [vp, C.c_int64]), ("fake_uint", [vp, C.c_uint64]),
                           ("fake_float", [vp, C.c_double]), ("fake_binary", [vp, C.c_char_p, C.c_size_t]),
                           ("fake_nil", [vp]), ("fake_cons", [vp, term, term]), ("fake_list", [vp, C.c_uint, C.POINTER(term)]),
                           ("fake_float_list", [vp, C.c_uint, C.POINTER(C.c_double)]),
                           ("fake_tuple", [vp, C.c_uint, C.POINTER(term)]), ("fake_copy_resource", [vp, term]),
                           ("fake_head", [term]), ("fake_tail", [term]), ("fake_tuple_element", [term, C.c_uint])):
            getattr(L, name).restype = term
            getattr(L, name).argtypes = args
        L.fake_kind.argtypes = [term]
        L.fake_atom_name.restype = C.c_char_p
        L.fake_atom_name.argtypes = [term]
        L.fake_int_value.restype = C.c_int64
        L.fake_int_value.argtypes = [term]
        L.fake_uint_value.restype = C.c_uint64
        L.fake_uint_value.argtypes = [term]
        L.fake_float_value.restype = C.c_double
        L.fake_float_value.argtypes = [term]
        L.fake_binary_size.restype = C.c_size_t
        L.fake_binary_size.argtypes = [term]
        L.fake_binary_data.restype = C.POINTER(C.c_ubyte)
        L.fake_binary_data.argtypes = [term]
        L.fake_tuple_arity.argtypes = [term]
        L.fake_live_resources.restype = C.c_long
        L.fake_dtor_calls.restype = C.c_long
        self.L = L
        self.nfuncs = L.fake_load()
        assert self.nfuncs > 0, "load callback failed"

    # ------------------------------------------------------------------ the module
    @property
    def module(self):
        return self.L.fake_module_name().decode()

    def functions(self):
        return {(self.L.fake_func_name(i).decode(), self.L.fake_func_arity(i)): self.L.fake_func_flags(i)
                for i in range(self.nfuncs)}

    def live_resources(self):
        return self.L.fake_live_resources()

    def dtor_calls(self):
        return self.L.fake_dtor_calls()

    # ------------------------------------------------------------------ terms
    def to_term(self, env, v):
        L = self.L
        if isinstance(v, Atom):
            return L.fake_atom(env, str(v).encode())
        if isinstance(v, Resource):
            return L.fake_copy_resource(env, v.term)
        if isinstance(v, bool):
            return L.fake_atom(env, b"true" if v else b"false")
        if isinstance(v, (bytes, bytearray)):
            return L.fake_binary(env, bytes(v), len(v))
        if isinstance(v, float):
            return L.fake_float(env, v)
        if isinstance(v, int):
            return L.fake_int(env, v) if v < 2 ** 63 else L.fake_uint(env, v)
        if isinstance(v, FloatList):
            arr = (C.c_double * len(v.values))(*v.values)
            return L.fake_float_list(env, len(v.values), arr)
        if isinstance(v, (list, tuple)):
            elems = (C.c_void_p * max(1, len(v)))(*[self.to_term(env, e) for e in v])
            return (L.fake_list if isinstance(v, list) else L.fake_tuple)(env, len(v), elems)
        if isinstance(v, ImproperList):
            t = self.to_term(env, v.tail)
            for e in reversed(v.items):
                t = L.fake_cons(env, self.to_term(env, e), t)
            return t
        raise TypeError(type(v))

    def from_term(self, t):
        L = self.L
        k = L.fake_kind(t)
        if k == T_ATOM:
            return Atom(L.fake_atom_name(t).decode())
        if k == T_INT:
            return int(L.fake_int_value(t))
        if k == T_UINT:
            return int(L.fake_uint_value(t))
        if k == T_FLOAT:
            return float(L.fake_float_value(t))
        if k == T_BINARY:
            n = L.fake_binary_size(t)
            return bytes(bytearray(L.fake_binary_data(t)[:n])) if n else b""
        if k == T_NIL:
            return []
        if k == T_CONS:
            out = []
            while L.fake_kind(t) == T_CONS:
                out.append(self.from_term(L.fake_head(t)))
                t = L.fake_tail(t)
            assert L.fake_kind(t) == T_NIL
            return out
        if k == T_TUPLE:
            return tuple(self.from_term(L.fake_tuple_element(t, i)) for i in range(L.fake_tuple_arity(t)))
        if k == T_RESOURCE:
            return Resource(self, t)
        if k == T_BADARG:
            raise ArgumentError()
        raise AssertionError("term kind %d" % k)

    def call(self, name, *args):
        env = self.L.fake_env_new()
        try:
            argv = (C.c_void_p * max(1, len(args)))(*[self.to_term(env, a) for a in args])
            r = self.L.fake_call(env, name.encode(), len(args), argv)
            if not r:
                raise AttributeError("%s/%d is not in the NIF table" % (name, len(args)))
            return self.from_term(r)
        finally:
            self.L.fake_env_free(env)


    def call_timed(self, name, *args):
        """call(), and the seconds the NIF itself took: the argument terms are built before the clock starts, the
        result is read after it has stopped (a probe's view of the shim, without this runtime's Python in it)."""
        import time
        env = self.L.fake_env_new()
        try:
            argv = (C.c_void_p * max(1, len(args)))(*[self.to_term(env, a) for a in args])
            t0 = time.perf_counter()
            r = self.L.fake_call(env, name.encode(), len(args), argv)
            dt = time.perf_counter() - t0
            if not r:
                raise AttributeError("%s/%d is not in the NIF table" % (name, len(args)))
            return self.from_term(r), dt
        finally:
            self.L.fake_env_free(env)


class FloatList:
    """[float] built in one call (long vectors)."""

    def __init__(self, values):
        self.values = [float(x) for x in values]


class ImproperList:
    def __init__(self, items, tail):
        self.items, self.tail = items, tail
