R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -o p -- python3 $R/bench.py > $OUT/default.log 2>&1
grep -a "^{\"" $OUT/default.log | tail -1 > $OUT/default.json
cd $R
python3 - <<'PY'
import csv
out='gpurun_out/prof3'
rows=list(csv.reader(open(out+'/default/p_kernel_stats.csv')))
with open(out+'/r05_default_kernel_stats.csv','w',newline='') as f:
    w=csv.writer(f)
    for r in rows:
        r[0]=r[0][:140]; w.writerow(r)
for r in csv.DictReader(open(out+'/default/p_kernel_stats.csv')):
    if 'vt::' in r['Name'] and float(r['TotalDurationNs'] if 'TotalDurationNs' in r else 0) >= 0:
        n=r['Name']
        if any(k in n for k in ('scan_topk_kernel<0, 3, 128, false, false, false>','shadow_scores_kernel<0, 8','prefix_multi','cosine_scan_multi','cosine_scan_kernel<320','hamming_dist_kernel','mfma_scores_kernel<8, false','hamming_topk_kernel<128')):
            print(n[:80], r['Calls'], r['AverageNs'])
PY
