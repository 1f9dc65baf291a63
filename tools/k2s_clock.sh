#!/bin/bash
# tools/k2s_clock.sh [rows] -- the clock the chip holds under each timing mode of tools/k2s_probe (GRBM_GUI_ACTIVE / 8 /
# duration per dispatch): is the pass short of its two standalone rates because something stalls, or because the
# chip runs the combination slower?  Prints one line per (kernel, debug mode) in dispatch order.
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROWS=${1:-6000000}
OUT=$R/gpurun_out/k2s_clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -o p -- $R/tools/k2s_probe $ROWS 768 256 > $OUT/probe.jsonl 2> $OUT/probe.err
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/**/p_counter_collection.csv', recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'scores_kernel' not in r['Kernel_Name']:
        continue
    d = rows.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'].split('(')[0][-60:], 'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
    d[r['Counter_Name']] = float(r['Counter_Value'])
# group consecutive dispatches of one kernel with similar duration (the probe runs 8 reps per mode)
prev = None
group = []
def flush(g):
    if not g: return
    ns = sorted(x['ns'] for x in g)[len(g) // 2]
    cyc = sorted(x.get('GRBM_GUI_ACTIVE', 0) / 8 for x in g)[len(g) // 2]
    busy = sorted(x.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for x in g)[len(g) // 2]
    print("%-62s n=%3d  %.3f ms  %.3e cycles  %.2f GHz  matrix pipe busy %.2f" % (g[0]['name'], len(g), ns / 1e6, cyc, cyc / ns, busy / 1024 / max(cyc, 1)))
for k, x in rows.items():
    key = x['name']
    if prev is not None and (key != prev or abs(x['ns'] - group[-1]['ns']) > 0.04 * group[-1]['ns']):
        flush(group); group = []
    group.append(x); prev = key
flush(group)
PY
