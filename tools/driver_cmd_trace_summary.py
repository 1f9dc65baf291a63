#!/usr/bin/env python3
"""tools/driver_cmd_trace_summary.py [<dir with p_kernel_trace.csv> [<the run's JSON line>]] -- what rocprofv3's kernel trace of
LITERALLY the driver's command (`rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5`,
tools/refresh_profiles.sh: driver_cmd) says about the headline kernel: the dispatches over the 10 M-row corpus in the order
they ran, by the time since the first of them, beside the line's own figures.  (Through round 5 the first ~1.2 s of scanning
-- the stretch after bench.py handed its 30-GB source tensor back to the driver -- ran 2.5 % slower per kernel, and the driver's
20 steps sat inside it; since r06 setup ends with a pause, --release-wait: the table should be flat from the first row on.)  Writes the text that is kept as
profiles/<round>_driver_cmd_trace_summary.txt to stdout.  No GPU needed: it reads the merged gpurun_out/prof/."""
import csv
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof", "driver_cmd")
line_path = sys.argv[2] if len(sys.argv) > 2 else d.rstrip("/") + ".json"
KERNEL = "scan_topk_kernel<0, 3, 128, false, false, false>"

rows = [r for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv"))) if KERNEL in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows]
big = [(t, ms) for t, ms in dur if ms > 3.0]          # the scans of the 10 M-row corpus (config 2's N = 1 M scans take 0.45 ms)
t0 = big[0][0]
big = [((t - t0) / 1e9, ms) for t, ms in big]
line = json.loads(open(line_path).read().strip().splitlines()[-1])

print("command: rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5   (literally the driver's command)")
print("the JSON line of that run:")
print("  config.release_wait_s: %s" % line.get("config", {}).get("release_wait_s"))
print("  value (W warm-up + %d timed steps): %.2f queries/s, %.4f ms per step; long_run %.2f queries/s over %d steps" % (
    line["steps"], line["value"], line["ms_per_step"], line["long_run"]["value"], line["long_run"]["steps"]))
print("  roofline.avg_launch_ms (HIP events on the library's stream over the timed steps) %.4f ms -> frac %.4f of 8 TB/s" % (
    line["roofline"]["avg_launch_ms"], line["roofline"]["frac"]))
print("rocprofv3's kernel trace of the same process, %s (grid 131072):" % KERNEL)
print("  %d dispatches in all -- the side legs share the kernel (config 2's N = 1 M scans, the callers' passes), so the --stats average of" % len(rows))
print("  this command (%.4f ms) is a mixture; the %d dispatches over the 10 M-row corpus, by seconds since the first of them:" % (
    sum(ms for _, ms in dur) / len(dur), len(big)))
edges = [0.0, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0, 3.0, 5.0, 1e9]
for lo, hi in zip(edges, edges[1:]):
    blk = [ms for t, ms in big if lo <= t < hi]
    if blk:
        print("    %5.2f - %-5s s: %4d dispatches, avg %.4f ms, min %.4f  (%.0f GB/s = %.3f of 8 TB/s)" % (
            lo, ("%.2f" % hi) if hi < 1e8 else "end", len(blk), statistics.mean(blk), min(blk), 30.72 / statistics.mean(blk) * 1e3,
            30.72 / statistics.mean(blk) * 1e3 / 8000.0))
early = [ms for t, ms in big if t < 1.0]
late = [ms for t, ms in big if t >= 1.5]
if early and late:
    print("  first second %.4f ms, from 1.5 s on %.4f ms (%+.1f %%); kernel time <= ms_per_step." % (
        statistics.mean(early), statistics.mean(late), (statistics.mean(early) / statistics.mean(late) - 1) * 100))
