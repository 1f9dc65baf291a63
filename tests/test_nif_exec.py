"""The erl_nif shim EXECUTED without Erlang/OTP and without a GPU: built
against the fake term runtime of tests/stubs/erl_nif_fake.c (tests/nif_runtime.py), loaded,
and driven through every decoding path that ends before the device is needed -- the module
load callback, the ErlNifFunc table, rustler-style decode failures (ArgumentError), the
{:error, binary} tuple of a compute call on a box with no HIP device.  The same runtime drives
the shim against the real library in tests/test_gpu_nif_exec.py (-m gpu).
"""
import re
import os

import pytest

import nif_runtime
from nif_runtime import ArgumentError, Atom, ImproperList

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rt():
    return nif_runtime.Runtime()


def test_module_loads_and_exposes_the_table(rt):
    assert rt.module == "Elixir.Vettore.Gpu.Nifs"
    funcs = rt.functions()
    ex = open(os.path.join(ROOT, "integration", "lib", "vettore", "gpu", "nifs.ex")).read()
    stubs = {(m.group(1), len([a for a in m.group(2).split(",") if a.strip()]))
             for m in re.finditer(r"def ([a-z0-9_]+)\(([^)]*)\), do: :erlang\.nif_error", ex)}
    assert set(funcs) == stubs and len(funcs) >= 14
    # nifs.rs marks every NIF schedule = "DirtyCpu"; the shim's are dirty jobs too (1 = CPU bound, 2 = IO bound)
    assert all(flags in (1, 2) for flags in funcs.values())
    with pytest.raises(AttributeError):
        rt.call("flat_search", 1)   # wrong arity: not in the table


def test_decode_failures_are_badarg(rt):
    """What rustler's decoders reject must come back as ArgumentError, before any device call."""
    # a reference that is not one
    for bad_ref in (1, b"x", [], Atom("nil"), (1, 2)):
        with pytest.raises(ArgumentError):
            rt.call("flat_search", bad_ref, [1.0], 1)
        with pytest.raises(ArgumentError):
            rt.call("flat_insert", bad_ref, b"a", [1.0])
        with pytest.raises(ArgumentError):
            rt.call("flat_delete", bad_ref, b"a")
    # Vec<f32>: integers are not floats (vettore_distance.ex:659-660 converts before the call),
    # atoms are not numbers, a double beyond f32's range does not fit, improper lists are not lists
    for bad in ([1], [1.0, 2], [Atom("nan")], [1e39], [-1e39], Atom("x"), b"\x00\x00\x80\x3f",
                ImproperList([1.0], 2.0), (1.0, 2.0)):
        with pytest.raises(ArgumentError):
            rt.call("normalize_l2", bad)
        with pytest.raises(ArgumentError):
            rt.call("compress_sign_bits", bad)
    # ragged batches: [{binary, [f32]}]
    for bad in (Atom("x"), [(b"a",)], [(b"a", [1.0], 3)], [(1, [1.0])], [(b"a", [1])], [[b"a", [1.0]]],
                ImproperList([(b"a", [1.0])], 7)):
        with pytest.raises(ArgumentError):
            rt.call("vector_top_k", bad, [1.0], 0, 1, 1)
    with pytest.raises(ArgumentError):
        rt.call("vector_top_k", [(b"a", [1.0])], [1.0], 0, -1, 1)          # usize
    with pytest.raises(ArgumentError):
        rt.call("vector_top_k", [(b"a", [1.0])], [1.0], 0, 1, 1.0)         # usize
    with pytest.raises(ArgumentError):
        rt.call("vector_top_k", [(b"a", [1.0])], [1.0], 2 ** 40, 1, 1)     # metric code: u8 in the reference, int here
    for bad in ([(b"a", [1.0])], [(b"a", [-1])], [(b"a", [Atom("x")])]):
        with pytest.raises(ArgumentError):
            rt.call("binary_top_k", bad, [1], 1, 1)
    with pytest.raises(ArgumentError):
        rt.call("binary_top_k", [(b"a", [1])], [1.5], 1, 1)
    # flat_new(metric_code, [device])
    for args in ((Atom("l2"), [0]), (0, []), (0, [Atom("gpu")]), (0, [-1]), (0, 0), (0, list(range(65)))):
        with pytest.raises(ArgumentError):
            rt.call("flat_new", *args)


def test_without_a_device_compute_calls_are_error_tuples_not_crashes(rt):
    """On this box (no HIP device) the library has no CPU fallback: the shim must hand its
    message over as {:error, binary} -- and must not leak a resource doing so."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "vettore_amd", "lib", "libvettore_hip.so"))
    if lib.vt_device_count() > 0:
        pytest.skip("a GPU is present: covered by tests/test_gpu_nif_exec.py")
    for call in (("flat_new", 0, [0]), ("normalize_l2", [3.0, 4.0]), ("compress_sign_bits", [1.0, -1.0, 0.0]),
                 ("vector_top_k", [(b"a", [1.0, 0.0])], [1.0, 0.0], 0, 2, 1), ("binary_top_k", [(b"a", [1])], [1], 1, 1)):
        tag, msg = rt.call(*call)
        assert tag == Atom("error") and isinstance(msg, bytes) and b"no HIP device" in msg, call
    assert rt.live_resources() == 0
