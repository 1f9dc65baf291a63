"""Shared helpers: fixture loading and the step-script runner used by both the
oracle tests (CPU) and the GPU parity tests (same scripts, same expectations)."""
import json
import math
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def b(x):
    return x.encode() if isinstance(x, str) else x


def close(actual, expected, tol):
    """assert_close of distances.rs:485-491: |a-e| <= tol*max(1,|a|,|e|)."""
    scale = max(1.0, abs(actual), abs(expected))
    return abs(actual - expected) <= tol * scale


def same_f32(a, b_):
    """bit equality of two f32 values (distinguishes -0.0 from 0.0)."""
    return np.float32(a).tobytes() == np.float32(b_).tobytes()


class IndexAdapter:
    """Uniform view of an index for run_steps: methods raise `error_type`
    carrying the reference's error string."""

    error_type = Exception

    def insert(self, id_, vector): ...
    def insert_many(self, items): ...
    def delete(self, id_): ...
    def search(self, query, limit): ...
    def __len__(self): ...
    dimension = None


def run_steps(index, steps, error_type):
    for i, st in enumerate(steps):
        op = st["op"]
        where = "step %d (%s)" % (i, op)
        try:
            if op == "insert":
                res = index.insert(st["id"], st["vector"])
            elif op == "insert_many":
                res = index.insert_many([(it[0], it[1]) for it in st["items"]])
            elif op == "delete":
                res = index.delete(st["id"])
            elif op == "search":
                res = index.search(st["query"], st["limit"])
            elif op == "expect_len":
                assert len(index) == st["len"], where
                continue
            elif op == "expect_dimension":
                assert index.dimension == st["dimension"], where
                continue
            else:
                raise AssertionError("unknown op " + op)
        except error_type as e:
            assert "expect_error" in st, "%s raised %r" % (where, e)
            if st["expect_error"] is not True:
                assert str(e) == st["expect_error"], where
            continue
        assert "expect_error" not in st, "%s should have failed with %r" % (where, st["expect_error"])
        if op != "search":
            continue
        hits = res
        if "expect" in st:
            assert [(h[0], h[1]) for h in hits] == [(b(e[0]), e[1]) for e in st["expect"]], where
        if "expect_ids" in st:
            assert [h[0] for h in hits] == [b(x) for x in st["expect_ids"]], where
        if "expect_first_id" in st:
            assert hits and hits[0][0] == b(st["expect_first_id"]), where
        if "expect_len" in st:
            assert len(hits) == st["expect_len"], where
        if st.get("expect_finite"):
            assert all(math.isfinite(h[1]) for h in hits), where
        if "expect_close" in st:
            for h, e in zip(hits, st["expect_close"]):
                assert h[0] == b(e[0]) and close(h[1], e[1], st["rel_tol"]), where


def total_key(x):
    """f32::total_cmp as a sortable int."""
    bits = int(np.float32(x).view(np.int32))
    return bits ^ ((bits >> 31) & 0x7FFFFFFF) if bits < 0 else bits


def full_sort(rows, raw_fn, rank_fn, limit):
    """The reference tests' differential oracle: score everything, sort by
    (rank.total_cmp, id bytes), truncate (flat.rs:222-241, search.rs:131-155)."""
    scored = [(b(i), raw_fn(v)) for i, v in rows]
    scored.sort(key=lambda t: (total_key(rank_fn(t[1])), t[0]))
    return scored[:limit]


def rerun_with_hooks_library(request):
    """For the tests that need the fault-injection hooks: the product library has none, so the
    test runs again in a child process whose vettore_amd loads lib/libvettore_hip_hooks.so (same
    sources, host side built with -DVT_TEST_HOOKS).  Returns True in the parent (the child's
    verdict has been asserted), False in the child (go on with the test body)."""
    import subprocess
    import sys
    if os.environ.get("VETTORE_HIP_LIB"):
        return False
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hooks = os.path.join(root, "vettore_amd", "lib", "libvettore_hip_hooks.so")
    assert os.path.exists(hooks), "build it with `make` (libvettore_hip_hooks.so)"
    env = dict(os.environ, VETTORE_HIP_LIB=hooks)
    res = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", request.node.nodeid],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    return True


def callers_meet(ref, queries, limit, kinds, params=None, candidates=None, rounds=4, hold=True):
    """Native threads that leave a barrier together, `rounds` times (tools/callers_native.cpp: vt_callers_meet), one
    thread per entry of `kinds` (0: flat_search, 1: quantized_search, 2: funnel_search over one stage).  With the hooks
    build loaded (rerun_with_hooks_library) and `hold`, the handle's first caller of a round keeps its slot until all the
    others have queued: who travels with whom is then a fact.  Every answer is compared inside with the same call made
    alone.  Returns (mismatches, failures)."""
    import ctypes as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hooks = "hooks" in os.path.basename(os.environ.get("VETTORE_HIP_LIB", ""))
    assert hooks or not hold, "callers are only held in libvettore_hip_hooks.so"
    path = os.path.join(root, "vettore_amd", "lib", "libvt_callers_hooks.so" if hooks else "libvt_callers.so")
    assert os.path.exists(path), "build it with `make` (%s)" % os.path.basename(path)
    lib = C.CDLL(path)
    lib.vt_callers_meet.restype = C.c_int
    lib.vt_callers_meet.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_int),
                                    C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    q = np.ascontiguousarray(queries, dtype=np.float32)
    t = len(kinds)
    kinds_a = (C.c_int * t)(*kinds)
    params_a = (C.c_size_t * t)(*(params or [0] * t))
    cands_a = (C.c_size_t * t)(*(candidates or [0] * t))
    wrong, failed = C.c_ulonglong(), C.c_ulonglong()
    rc = lib.vt_callers_meet(ref.handle, q.ctypes.data_as(C.POINTER(C.c_float)), q.shape[0], q.shape[1], limit, kinds_a,
                             params_a, cands_a, t, rounds, 1 if hold else 0, C.byref(wrong), C.byref(failed))
    assert rc == 0, "vt_callers_meet: %d" % rc
    return int(wrong.value), int(failed.value)
