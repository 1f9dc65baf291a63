R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for name in single default; do
  if [ $name = single ]; then ARGS="--no-side"; else ARGS=""; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o p -- python3 $R/bench.py $ARGS > $OUT/$name.log 2>&1
  tail -1 $OUT/$name.log > $OUT/$name.json
done
cd $R
python3 - <<'PY'
import csv
out='gpurun_out/prof'
for name in ('single','default'):
    rows=list(csv.reader(open('%s/%s/p_kernel_stats.csv'%(out,name))))
    with open('%s/r02_%s_kernel_stats.csv'%(out,name),'w',newline='') as f:
        w=csv.writer(f)
        for r in rows:
            r[0]=r[0][:140]; w.writerow(r)
    for r in csv.DictReader(open('%s/%s/p_kernel_stats.csv'%(out,name))):
        if 'vt::' in r['Name']: print(name, r['Name'][:70], r['Calls'], r['AverageNs'])
    print(open('%s/%s.json'%(out,name)).read()[:400])
PY
