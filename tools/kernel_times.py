#!/usr/bin/env python3
"""tools/kernel_times.py DIR [substring] -- per-kernel launch count / average / minimum duration (us)
from a `rocprofv3 --kernel-trace` output directory, whichever format it wrote (csv or rocpd sqlite)."""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def rows(d):
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        kd = [t for t in tabs if "kernel_dispatch" in t]
        sym = [t for t in tabs if "kernel_symbol" in t]
        if not kd or not sym:
            continue
        q = f"select s.kernel_name, d.end - d.start from {kd[0]} d join {sym[0]} s on d.kernel_id = s.id"
        yield from db.execute(q)


def main():
    d = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(list)
    for name, ns in rows(d):
        if want in name:
            acc[name].append(ns)
    for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print(f"{len(v):6d} {sum(v) / len(v) / 1e3:10.1f} {min(v) / 1e3:10.1f}  {name[:110]}")


if __name__ == "__main__":
    main()
