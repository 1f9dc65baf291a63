R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/qtrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $R/bench.py --mode quantized --steps 300 --warmup 20 --no-cpu > $OUT/log 2>&1
cd $R
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/qtrace/p_kernel_stats.csv')):
    if 'vt::' in r['Name']: print(r['Name'][:70], r['Calls'], r['AverageNs'])
rows=list(csv.DictReader(open('gpurun_out/qtrace/p_kernel_trace.csv')))
ks=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:40]) for r in rows)
prev=None
for st,en,name in ks[-16:]:
    print("%-42s dur %7.1f gap %6.1f"%(name,(en-st)/1000,(st-prev)/1000 if prev else 0)); prev=en
PY
