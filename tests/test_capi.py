"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports
every symbol include/vettore_flat.h declares, reports the reference's error
strings, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vettore_flat.h")


@pytest.fixture(scope="module")
def lib():
    import vettore_amd._lib as L
    return L.load()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), "libvettore_hip.so does not export " + name


def test_nothing_but_the_header_is_exported():
    """... and the other direction (r06, csrc/exports.map): the dynamic symbol table of libvettore_hip.so -- and of the hooks
    build -- holds the C ABI and nothing else.  No kernel's host stub, no launcher of csrc/vt_device.h, no instantiation of the
    standard library can collide with, or be interposed by, whatever else lives in the process that loads it (a BEAM)."""
    import subprocess
    declared = set(declared_functions())
    for name in ("libvettore_hip.so", "libvettore_hip_hooks.so"):
        so = os.path.join(ROOT, "vettore_amd", "lib", name)
        if not os.path.exists(so):
            continue
        out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
        exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
        assert exported == declared, (name, sorted(exported - declared)[:5], sorted(declared - exported)[:5])


def test_python_binding_covers_the_header():
    import vettore_amd._lib as L
    assert sorted(L.SYMBOLS) == declared_functions()


def test_error_strings_are_the_references(lib):
    # native/vettore/src/flat.rs:138,141; distances.rs:36,67,135,463; search.rs:47
    want = {
        1: "vector must not be empty",
        2: "dimension mismatch",
        3: "vector contains a non-finite value",
        4: "metric overflow",
        5: "unknown metric",
        6: "invalid prefix dimensions",
        7: "dimensions must be positive",
        8: "flat lock poisoned",            # nifs.rs:269
    }
    for code, text in want.items():
        assert lib.vt_strerror(code).decode() == text
    assert lib.vt_abi_version() == 4


def test_header_cites_the_reference_interface():
    text = open(HEADER).read()
    for cite in ("nifs.rs:259-271", "nifs.rs:273-284", "nifs.rs:286-295", "nifs.rs:297-309", "nifs.rs:151-162",
                 "nifs.rs:164-175", "nifs.rs:107-111", "nifs.rs:125-129", "flat.rs:96-124"):
        assert cite in text, cite


def test_argument_validation_happens_before_the_device(lib):
    """Statuses that need no GPU: metric decode and helper argument checks
    follow the reference's order (nifs.rs:158-161, search.rs:46-48, :82-84)."""
    h = C.c_void_p()
    assert lib.vt_flat_new(9, 0, C.byref(h)) == 5          # unknown metric
    assert lib.vt_flat_new(-1, 0, C.byref(h)) == 5
    devs = (C.c_int * 2)(0, 1)
    assert lib.vt_flat_new_sharded(9, devs, 2, C.byref(h)) == 5
    assert lib.vt_flat_new_sharded(2, devs, 0, C.byref(h)) == 19   # no devices: VT_ERR_ARGUMENT
    one = (C.c_float * 1)(1.0)
    off = (C.c_size_t * 1)(0)
    out = C.c_void_p()
    assert lib.vt_vector_top_k(0, 0, b"", off, one, off, one, 1, 9, 1, 1, C.byref(out)) == 5
    assert lib.vt_vector_top_k(0, 0, b"", off, one, off, one, 1, 0, 0, 1, C.byref(out)) == 6   # prefix 0
    assert lib.vt_vector_top_k(0, 0, b"", off, one, off, one, 1, 0, 2, 1, C.byref(out)) == 6   # prefix > len
    nan = (C.c_float * 1)(float("nan"))
    assert lib.vt_vector_top_k(0, 0, b"", off, one, off, nan, 1, 0, 1, 1, C.byref(out)) == 3
    q = (C.c_uint64 * 1)(0)
    assert lib.vt_binary_top_k(0, 0, b"", off, q, off, q, 0, 0, 1, C.byref(out)) == 7         # dims == 0
    assert lib.vt_binary_top_k(0, 0, b"", off, q, off, q, 0, 1, 1, C.byref(out)) == 2         # no query words
    assert lib.vt_binary_top_k(0, 0, b"", off, q, off, q, 1, 1, 1, C.byref(out)) == 0         # empty batch ok
    assert lib.vt_hits_len(out) == 0
    lib.vt_hits_free(out)
    assert lib.vt_normalize_l2(0, 1, 1, nan, one) == 3


def test_no_cpu_fallback_without_a_device(lib):
    if lib.vt_device_count() > 0:
        pytest.skip("a HIP device is present")
    h = C.c_void_p()
    assert lib.vt_flat_new(2, 0, C.byref(h)) == 17         # VT_ERR_DEVICE
    assert b"no CPU fallback" in lib.vt_last_error()
    devs = (C.c_int * 2)(0, 1)
    assert lib.vt_flat_new_sharded(2, devs, 2, C.byref(h)) == 17
    from vettore_amd import nifs
    with pytest.raises(RuntimeError, match="device error"):
        nifs.flat_new_cosine()


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under vettore_amd/ may import,
    link or load it."""
    pkg = os.path.join(ROOT, "vettore_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r'#include\s*[<"][^>"]*oracle|libvt_oracle|^\s*(import|from)\s+oracle', text, re.M), f
    import subprocess
    so = os.path.join(pkg, "lib", "libvettore_hip.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "oracle" not in needed


def test_product_library_carries_no_test_hooks():
    """The fault-injection switches (test_* settings, VT_TEST_* variables) exist in libvettore_hip_hooks.so only, and
    the switches that force a path on corpora the cost model would never send there have no environment name at all:
    the product library has no environment variable that makes it fail or take a detour on purpose."""
    lib_dir = os.path.join(ROOT, "vettore_amd", "lib")
    product = open(os.path.join(lib_dir, "libvettore_hip.so"), "rb").read()
    assert b"VT_TEST_" not in product and b"VT_FORCE_" not in product and b"test_fail_after_id_update" not in product
    # (r06) ... nor the timing experiments' switches, nor a kernel of a path that lost its A/B
    for gone in (b"batch_debug", b"mq_dbg", b"trace_batch", b"hybrid_chain", b"test_coalesce_hold_until", b"union_rows_kernel",
                 b"mfma_scores_kernel3"):
        assert gone not in product, gone
    hooks = os.path.join(lib_dir, "libvettore_hip_hooks.so")
    if os.path.exists(hooks):
        data = open(hooks, "rb").read()
        for name in (b"test_fail_after_id_update", b"test_exchange_stall_ms", b"test_refuse_nzbits"):
            assert name in data, name


def test_settings_by_name(lib):
    """vt_debug_set / vt_debug_get (include/vettore_flat.h): the library's switches after the one read of the
    environment at load time.  No device needed."""
    v = C.c_long(-1)
    assert lib.vt_debug_get(b"bf16_min_rank", C.byref(v)) == 0 and v.value == 6
    assert lib.vt_debug_get(b"force_batch_mfma", C.byref(v)) == 0 and v.value == 0
    assert lib.vt_debug_set(b"force_batch_mfma", 1) == 0
    assert lib.vt_debug_get(b"force_batch_mfma", C.byref(v)) == 0 and v.value == 1
    assert lib.vt_debug_set(b"force_batch_mfma", 0) == 0
    for bad in (b"VT_FORCE_BATCH_MFMA", b"no_such_switch", b"test_refuse_shadow", b""):
        assert lib.vt_debug_set(bad, 1) == 19, bad      # VT_ERR_ARGUMENT
    assert lib.vt_debug_set(None, 1) == 19 and lib.vt_debug_get(b"coalesce", None) == 19
    # r06: the A/B switches whose alternative lost are gone with their paths, the timing experiments live in a build of
    # their own (make experiments) ...
    for gone in (b"hybrid_chain", b"direct_query", b"batch_kernel", b"pm_panel", b"cs_panel", b"scan_rt_order", b"multi_general",
                 b"ingest_serial", b"ingest_separate_check", b"funnel_dense_sample", b"hamming_lists", b"batch_pass_five",
                 b"rescore_blocks", b"no_pattern_bits", b"batch_debug", b"mq_dbg", b"trace_batch", b"test_coalesce_hold_until"):
        assert lib.vt_debug_set(gone, 1) == 19 and lib.vt_debug_get(gone, C.byref(v)) == 19, gone
    # ... and a value the setting's own parser could not have produced is refused (reduce_order = 7 would hand
    # new indexes a lane order no kernel has)
    for name, bad_values, good in ((b"reduce_order", (-1, 4, 7), 3), (b"batch_nominate", (0, 3), 2), (b"batch_shadow", (2, -1), 1),
                                   (b"slab", (2,), 0), (b"shard_exchange", (3, -1), 0)):
        for bad in bad_values:
            assert lib.vt_debug_set(name, bad) == 19, (name, bad)
        assert lib.vt_debug_get(name, C.byref(v)) == 0 and v.value == good, (name, v.value)
        assert lib.vt_debug_set(name, good) == 0
    # the environment is not consulted again: a variable set now changes nothing
    os.environ["VT_COALESCE_SLOTS"] = "5"
    try:
        assert lib.vt_debug_get(b"coalesce_slots", C.byref(v)) == 0 and v.value == 0
    finally:
        del os.environ["VT_COALESCE_SLOTS"]


def test_the_environment_is_read_when_the_library_is_loaded():
    """... and only then: a fresh process with VT_* set sees them in the table (string-valued ones as their codes)."""
    import subprocess
    import sys
    code = ("import ctypes as C, vettore_amd._lib as L; l = L.load(); v = C.c_long()\n"
            "out = []\n"
            "for n in (b'reduce_order', b'batch_nominate', b'batch_shadow', b'slab', b'shard_exchange', b'coalesce', b'no_multi_scan', b'ingest_stage_mb'):\n"
            "    assert l.vt_debug_get(n, C.byref(v)) == 0; out.append(v.value)\n"
            "print(out)")
    env = dict(os.environ, VT_REDUCE_ORDER="avx", VT_BATCH_NOMINATE="f32", VT_BATCH_SHADOW="off", VT_SLAB="malloc",
               VT_SHARD_EXCHANGE="rccl", VT_COALESCE="0", VT_NO_MULTI_SCAN="1", VT_INGEST_STAGE_MB="16")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "[1, 1, 0, 1, 2, 0, 1, 16]", (r.stdout, r.stderr[-2000:])
