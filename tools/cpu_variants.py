#!/usr/bin/env python3
"""CPU baseline variants beside the GPU numbers: the oracle's
reference-shaped search (hash map of separately allocated rows, per-row id clone,
bounded heap -- flat.rs:96-124) and its contiguous-matrix search ("best-effort CPU"),
each built -O3 for baseline x86-64 (what a precompiled NIF targets) and with
-march=native (Taskfile.yml:12), on one thread and on T concurrent readers.
Prints one JSON line per variant.  Test/bench infrastructure only."""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "oracle", "_build")
FLAGS = {"x86-64": ["-O3"], "native": ["-O3", "-march=native"]}


def build(tag):
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "libvt_oracle_%s.so" % tag)
    cmd = ["gcc", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math", *FLAGS[tag], "-shared", "-o", out,
           os.path.join(ROOT, "oracle", "vt_oracle.c"), "-lm"]
    subprocess.check_call(cmd)
    return out


def child(tag, rows, dim, seconds):
    import oracle
    rng = np.random.default_rng(20260721)
    x = np.empty((rows, dim), dtype=np.float32)
    for s0 in range(0, rows, 1 << 16):   # (in pieces: a sample of several GB must not exist twice, let alone in f64)
        blk = rng.random((min(rows, s0 + (1 << 16)) - s0, dim), dtype=np.float32) * 2.0 - 1.0
        blk /= np.sqrt(np.sum(blk.astype(np.float64) ** 2, axis=1, keepdims=True)).astype(np.float32)
        x[s0:s0 + len(blk)] = blk
    ids = [b"doc-%d" % (i + 1) for i in range(rows)]
    packed = oracle.pack_ids(ids)
    ix = oracle.FlatIndex(2)
    ix.insert_matrix(ids, x)
    qs = np.random.default_rng(20260722).uniform(-1, 1, size=(64, dim)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    shapes = {"reference-shaped (hash map of rows, id clone per row)": lambda q: ix.search(q, 10),
              "contiguous matrix": lambda q: oracle.matrix_search(2, x, packed, q, 10)}
    T = max(1, int(os.environ.get("THREADS", "0")) or os.cpu_count() or 1)
    for shape, fn in shapes.items():
        for threads in (1, T):
            counts = [0] * threads
            stop = time.perf_counter() + seconds

            def reader(t):
                i = t
                while time.perf_counter() < stop:
                    fn(qs[i % len(qs)])
                    counts[t] += 1
                    i += 1

            fn(qs[0])
            t0 = time.perf_counter()
            ths = [threading.Thread(target=reader, args=(t,)) for t in range(threads)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dt = time.perf_counter() - t0
            rps = rows * sum(counts) / dt
            print(json.dumps({"build": tag, "shape": shape, "threads": threads, "sample_rows": rows, "dim": dim,
                              "rows_per_s": round(rps), "effective_GBps": round(rps * dim * 4 / 1e9, 2),
                              "queries_per_s_at_10M": round(rps / 1e7, 3)}), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
        return
    rows = int(os.environ.get("ROWS", "200000"))
    seconds = float(os.environ.get("SECONDS_PER_LEG", "4"))
    for tag in FLAGS:
        env = dict(os.environ, VT_ORACLE_LIB=build(tag))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", tag, str(rows), "768", str(seconds)],
                              env=env)


if __name__ == "__main__":
    main()
