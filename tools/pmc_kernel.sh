#!/bin/bash
# tools/pmc_kernel.sh KERNEL_SUBSTRING OUT_TXT -- bench.py args...
# rocprofv3 PMC passes (one counter set per run, --kernel-trace only: what the pool allows) over a
# bench.py invocation; prints and writes per-launch averages of the kernels whose name contains
# KERNEL_SUBSTRING, with the derived figures DESIGN.md quotes (clock, matrix-pipe busy, LDS busy,
# parked / stalled shares, HBM bytes).  SQ counters are sums over waves / CUs, GRBM_GUI_ACTIVE over 8 XCDs;
# FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: gfx950 tallies 128-B requests at 64 B).
KSUB=$1; OUTTXT=$2; shift 2
[ "$1" = "--" ] && shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$(echo "$KSUB" | tr -c 'A-Za-z0-9' '_' | cut -c1-24)
OUT=$R/gpurun_out/pmck/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -o p -- python3 $R/bench.py "$@" > $OUT/set$i.log 2>&1
done
cd $R
python3 - "$KSUB" "$OUT" "$OUTTXT" "$*" <<'PY'
import csv, glob, collections, sys
ksub, out, outtxt, cmd = sys.argv[1:5]
acc = collections.defaultdict(list)
dur = []
names = set()
for f in sorted(glob.glob(out + '/set*/**/p_counter_collection.csv', recursive=True)):
    seen = {}
    for r in csv.DictReader(open(f)):
        if ksub in r['Kernel_Name']:
            names.add(r['Kernel_Name'][:120])
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            seen[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    dur += list(seen.values())
avg = {k: sum(v) / len(v) for k, v in acc.items()}
lines = ["kernel(s): " + "; ".join(sorted(names)),
         "command: rocprofv3 --pmc <one set per pass> --kernel-trace -- python3 bench.py " + cmd,
         "per launch (averages over %d profiled launches; duration under the profiler %.3f ms):" % (len(dur), sum(dur) / max(1, len(dur)) / 1e6)]
for k in sorted(avg):
    lines.append("  %-26s %.4g" % (k, avg[k]))
d_ms = sum(dur) / max(1, len(dur)) / 1e6
if 'GRBM_GUI_ACTIVE' in avg and d_ms:
    cyc = avg['GRBM_GUI_ACTIVE'] / 8
    ghz = cyc / (d_ms * 1e6)
    lines.append("derived: kernel cycles %.4g (= %.2f GHz over %.3f ms)" % (cyc, ghz, d_ms))
    simds = 256 * 4
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg:
        lines.append("  matrix pipe busy per SIMD            %.3f" % (avg['SQ_VALU_MFMA_BUSY_CYCLES'] / simds / cyc))
    if 'SQ_LDS_IDX_ACTIVE' in avg:
        lines.append("  LDS array busy per CU                %.3f   (bank conflicts %.4f of LDS cycles)" % (
            avg['SQ_LDS_IDX_ACTIVE'] / 256 / cyc, avg.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, avg['SQ_LDS_IDX_ACTIVE'])))
    if 'SQ_WAVE_CYCLES' in avg:
        wc = avg['SQ_WAVE_CYCLES']
        lines.append("  waves parked (s_waitcnt / s_barrier) %.3f of their cycles, stalled at issue %.3f" % (
            avg.get('SQ_WAIT_ANY', 0) / wc, avg.get('SQ_WAIT_INST_ANY', 0) / wc))
        lines.append("  resident waves                       %.0f (SQ_WAVE_CYCLES x 4 / kernel cycles)" % (wc * 4 / cyc))
    if 'SQ_INSTS_MFMA' in avg:
        lines.append("  instructions per MFMA: VALU %.2f, LDS %.3f, SALU %.2f, VMEM %.3f" % tuple(
            avg.get(k, 0) / avg['SQ_INSTS_MFMA'] for k in ('SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM')))
if 'FETCH_SIZE' in avg:
    lines.append("  HBM read bytes per launch            %.5g  (FETCH_SIZE KiB x 1024 x 2)" % (avg['FETCH_SIZE'] * 1024 * 2))
if 'WRITE_SIZE' in avg:
    lines.append("  HBM written bytes per launch         %.5g  (WRITE_SIZE KiB x 1024)" % (avg['WRITE_SIZE'] * 1024))
txt = "\n".join(lines) + "\n"
open(outtxt, 'w').write(txt)
print(txt)
PY
