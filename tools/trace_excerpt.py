#!/usr/bin/env python3
"""tools/trace_excerpt.py KERNEL_TRACE_CSV [ANCHOR [MIN_US]] -- three consecutive 256-query groups of one vt_flat_search_batch call out of a
rocprofv3 --kernel-trace of `bench.py --mode batch --batch 4096`: every dispatch with start / end (ms from the first
group's pass) and its queue, so that one can read off which kernels of group g run inside group g + 1's pass over the
rows (the exact rescoring scan_topk_kernel and batch_select_kernel of the previous group, the sample pass and
sample_tau_groups_kernel of the next), then the gaps between consecutive passes over the whole trace.
ANCHOR (default "shadow_scores_kernel<0") names the kernel that counts as a group's pass, MIN_US (default 2000) how long a
launch of it must be to count -- `cosine_scan_multi_kernel 500` cuts a funnel batch's groups of eight the same way,
`hamming_dist_multi_kernel 100` a quantized batch's (tools/group_pipeline_probe.py under rocprofv3 --kernel-trace)."""
import csv
import re
import sys


def short(name):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or ""))[:48] if m else name[:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], r["Grid_Size_X"],
                 r["LDS_Block_Size"]) for r in rows)
    anchor = sys.argv[2] if len(sys.argv) > 2 else "shadow_scores_kernel<0"
    min_ns = int(float(sys.argv[3]) * 1000) if len(sys.argv) > 3 else 2_000_000
    passes = [i for i, e in enumerate(ev) if e[2].startswith(anchor) and e[1] - e[0] > min_ns]
    if len(passes) < 8:
        print("fewer than eight passes in the trace")
        return
    first = passes[-8]
    t0 = ev[first][0]
    print("# dispatches around three consecutive groups (ms from the first pass's start; q = HSA queue = the group's stream)")
    print("#   start      end   length  queue  kernel                                            grid      LDS")
    for e in ev[max(0, first - 8):passes[-5] + 1]:
        print("%9.3f %9.3f %8.3f  q%-4s %-48s %8s %8s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[3], e[2], e[4], e[5]))
    p = [ev[i] for i in passes]
    gaps = [(p[i + 1][0] - p[i][1]) / 1e6 for i in range(len(p) - 1)]
    inside = [g for g in gaps if g < 1.0]
    print("# %d passes; gap between the end of a pass and the start of the next (ms), calls' own boundaries left out:" % len(p))
    print("#   median %.3f, mean %.3f, max %.3f over %d gaps" % (sorted(inside)[len(inside) // 2], sum(inside) / len(inside), max(inside), len(inside)))
    durs = sorted((e[1] - e[0]) / 1e6 for e in p)
    print("#   pass length (ms): median %.3f, min %.3f, max %.3f" % (durs[len(durs) // 2], durs[0], durs[-1]))
    others = ("scan_topk_kernel", "batch_select_kernel", "sample_tau_groups_kernel", "shadow_scores_kernel<2")
    if len(sys.argv) > 2:
        others = sorted({e[2].split("<")[0] for e in ev if "kernel" in e[2] and not e[2].startswith(anchor.split("<")[0])})
    for name in others:
        d = sorted((e[1] - e[0]) / 1e3 for e in ev if e[2].startswith(name))
        if d:
            print("#   %-28s %4d launches, median %.1f us, max %.1f us" % (name, len(d), d[len(d) // 2], d[-1]))


if __name__ == "__main__":
    main()
