#!/bin/bash
# PMC passes over bench.py --mode batch (one counter set per run); prints per-launch values
# of the full-pass MFMA kernel.  VT_BATCH_KERNEL selects the variant.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ROWS=${ROWS:-10000000}
for set in "GRBM_GUI_ACTIVE" "MfmaUtil" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcb/$tag -o p -- python3 $R/bench.py --mode batch --rows $ROWS --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmcb/*/p_counter_collection.csv')):
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if 'mfma_scores' in r['Kernel_Name'] and int(r['Grid_Size']) > 40000:
            by[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
            by[r['Dispatch_Id']]['dur_ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    for d, v in by.items():
        print(d, v)
PY
