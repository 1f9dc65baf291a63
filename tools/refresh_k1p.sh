R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export METRICS=0 PREFIXES=128 K1M=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefix_multi -o p -- python3 $R/tools/prefix_multi_probe.py > $OUT/prefix_multi.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prefix_multi_fetch -o p -- python3 $R/tools/prefix_multi_probe.py > $OUT/prefix_multi_fetch.log 2>&1
unset METRICS PREFIXES K1M
python3 $R/tools/prefix_multi_probe.py 2>/dev/null | grep "^{" > $OUT/r04_prefix_multi_probe.jsonl
ROWS=5000000 DIM=384 PREFIXES=128,384 python3 $R/tools/prefix_multi_probe.py 2>/dev/null | grep "^{" >> $OUT/r04_prefix_multi_probe.jsonl
cd $R
python3 - <<'PY'
import csv, os
out='gpurun_out/prof2'
rows=list(csv.reader(open(out+'/prefix_multi/p_kernel_stats.csv')))
with open(out+'/r04_prefix_multi_kernel_stats.csv','w',newline='') as f:
    w=csv.writer(f)
    for r in rows:
        r[0]=r[0][:140]; w.writerow(r)
rows=list(csv.DictReader(open(out+'/prefix_multi_fetch/p_counter_collection.csv')))
with open(out+'/r04_prefix_multi_pmc_fetch.csv','w',newline='') as f:
    w=csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows:
        if 'prefix_multi_kernel' in r['Kernel_Name']:
            r['Kernel_Name']=r['Kernel_Name'][:140]; w.writerow(r)
vals=[float(r['Counter_Value']) for r in rows if 'prefix_multi_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
big=[v for v in vals if v>1e6]
print("K1p sweeps", len(big), "FETCH x2 GB", sum(big)/len(big)*2*1024/1e9)
for r in csv.DictReader(open(out+'/prefix_multi/p_kernel_stats.csv')):
    if 'prefix_multi' in r['Name'] or 'sample_tau' in r['Name'] or 'scan_topk_kernel<1, 3, 320' in r['Name']:
        print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
cat $OUT/r04_prefix_multi_probe.jsonl | cut -c1-900
