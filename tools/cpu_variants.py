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


def child(tags, rows, dim, seconds):
    """Both builds in ONE process over ONE sample (the sample costs more to make than to scan).  The
    sample is cut into 8 parts, each an index of its own; a reader walks the parts round-robin (every
    reader from another start) and stops when its time is up, so a leg takes `seconds` plus at most
    one part -- 256 readers each finishing a whole 4-GB scan would be a minute per leg -- while the
    working set stays the whole sample."""
    import gc
    import oracle
    rng = np.random.default_rng(20260721)
    x = np.empty((rows, dim), dtype=np.float32)
    for s0 in range(0, rows, 1 << 16):   # (in pieces: a sample of several GB must not exist twice, let alone in f64)
        blk = rng.random((min(rows, s0 + (1 << 16)) - s0, dim), dtype=np.float32) * 2.0 - 1.0
        blk /= np.sqrt(np.einsum("ij,ij->i", blk, blk, dtype=np.float64))[:, None].astype(np.float32)
        x[s0:s0 + len(blk)] = blk
    ids = [b"doc-%d" % (i + 1) for i in range(rows)]
    qs = np.random.default_rng(20260722).uniform(-1, 1, size=(64, dim)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    P = 8
    cut = [rows * p // P for p in range(P + 1)]
    T = max(1, int(os.environ.get("THREADS", "0")) or os.cpu_count() or 1)
    for tag in tags:
        os.environ["VT_ORACLE_LIB"] = build(tag)
        oracle._lib = None   # bind this build (every object of the previous one is gone by now)
        oracle.lib()
        packed = [oracle.pack_ids(ids[cut[p]:cut[p + 1]]) for p in range(P)]
        parts = []
        for p in range(P):
            ix = oracle.FlatIndex(2)
            ix.insert_matrix(ids[cut[p]:cut[p + 1]], x[cut[p]:cut[p + 1]])
            parts.append(ix)
        shapes = {"reference-shaped (hash map of rows, id clone per row)": lambda q, p: parts[p].search(q, 10),
                  "contiguous matrix": lambda q, p: oracle.matrix_search(2, x[cut[p]:cut[p + 1]], packed[p], q, 10)}
        for shape, fn in shapes.items():
            for threads in (1, T):
                scanned = [0] * threads
                fn(qs[0], 0)
                stop = time.perf_counter() + seconds

                def reader(t):
                    i = t
                    while time.perf_counter() < stop:
                        p = i % P
                        fn(qs[(i // P) % len(qs)], p)
                        scanned[t] += cut[p + 1] - cut[p]
                        i += 1

                t0 = time.perf_counter()
                ths = [threading.Thread(target=reader, args=(t,)) for t in range(threads)]
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
                dt = time.perf_counter() - t0
                rps = sum(scanned) / dt
                print(json.dumps({"build": tag, "shape": shape, "threads": threads, "sample_rows": rows, "dim": dim,
                                  "rows_per_s": round(rps), "effective_GBps": round(rps * dim * 4 / 1e9, 2),
                                  "queries_per_s_at_10M": round(rps / 1e7, 3), "leg_seconds": round(dt, 2)}), flush=True)
        del parts, shapes, fn, ix
        gc.collect()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2].split(","), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
        return
    rows = int(os.environ.get("ROWS", "200000"))
    seconds = float(os.environ.get("SECONDS_PER_LEG", "4"))
    for tag in FLAGS:
        build(tag)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", ",".join(FLAGS), str(rows), "768", str(seconds)])


if __name__ == "__main__":
    main()
