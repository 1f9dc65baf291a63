#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ on a GPU box (run from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh'
# Kernel-trace/stats runs and PMC runs are separate passes, as the pool requires.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o p -- python3 $R/bench.py "$@" > $OUT/$name.log 2>&1
  grep -a "^{\"" $OUT/$name.log | tail -1 > $OUT/$name.json   # (rocprofv3 prints its own lines after the program's last)
}
pmc() {  # name, counter, bench args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$name -o p -- python3 $R/bench.py "$@" > $OUT/$name.log 2>&1
}
export RND=${RND:-r05}
stats single --no-side          # the headline alone: the scan kernel's average is the bench line's avg_launch_ms
stats default                   # the driver's command: headline + side legs (config 2 shares the scan kernel: 1 100 launches at N=1M)
pmc single_fetch FETCH_SIZE --steps 20 --warmup 2 --no-cpu --no-side
pmc single_write WRITE_SIZE --steps 20 --warmup 2 --no-cpu --no-side
stats batch --mode batch --nominate f32 --steps 6 --warmup 1 --no-cpu          # K2: FP32 matrix cores
stats batch_bf16 --mode batch --nominate bf16 --steps 24 --warmup 3 --no-cpu   # K2s: bf16 nomination from the bf16 shadow (the default since r04)
# config 3 as BASELINE.json writes it: 16 x 256 in ONE call (r05: groups alternate between two contexts) -- stats, and the
# kernel trace the excerpt below is cut from (group g's rescoring / select inside group g + 1's pass)
stats batch16 --mode batch --nominate bf16 --batch 4096 --steps 4 --warmup 1 --no-cpu
rocprofv3 --kernel-trace --output-format csv -d $OUT/batch16_trace -o t -- python3 $R/bench.py --mode batch --nominate bf16 --batch 4096 --steps 2 --warmup 1 --no-cpu > $OUT/batch16_trace.log 2>&1
stats batch_k2b --mode batch --nominate bf16 --shadow off --steps 12 --warmup 3 --no-cpu   # K2b: the same pass streaming the f32 rows (no shadow)
stats quantized --mode quantized --steps 300 --warmup 20 --no-cpu
pmc quantized_fetch FETCH_SIZE --mode quantized --steps 20 --warmup 2 --no-cpu
stats funnel --mode funnel --steps 200 --warmup 5 --no-cpu
pmc funnel_fetch FETCH_SIZE --mode funnel --steps 20 --warmup 2 --no-cpu
pmc batch_fetch FETCH_SIZE --mode batch --nominate f32 --steps 2 --warmup 1 --no-cpu
pmc batch_bf16_fetch FETCH_SIZE --mode batch --nominate bf16 --steps 4 --warmup 1 --no-cpu
pmc batch_k2b_fetch FETCH_SIZE --mode batch --nominate bf16 --shadow off --steps 4 --warmup 1 --no-cpu
# float hamming alone (K4 over the non-zero-bit column): the pass bench.py's side.pattern_hamming prices
ROWS=10000000 METRICS=7 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pattern_hamming -o p -- python3 $R/tools/pattern_probe.py > $OUT/pattern_hamming.log 2>&1
# K1m: 8 queries per sweep (manhattan, N=10M, d=768); the program after `--` is python3 itself
export ROWS=10000000 NQS=8 METRICS=5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/multi -o p -- python3 $R/tools/multi_probe.py > $OUT/multi.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/multi_fetch -o p -- python3 $R/tools/multi_probe.py > $OUT/multi_fetch.log 2>&1
unset ROWS NQS METRICS
# K1p: eight L2 funnel searches per sweep of the 128-float prefixes (N=10M, d=768) -- stats pass, FETCH pass
export METRICS=0 PREFIXES=128 K1M=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefix_multi -o p -- python3 $R/tools/prefix_multi_probe.py > $OUT/prefix_multi.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prefix_multi_fetch -o p -- python3 $R/tools/prefix_multi_probe.py > $OUT/prefix_multi_fetch.log 2>&1
unset METRICS PREFIXES K1M
cd $R
python3 - <<'PY'
import csv, glob, json, os
out = 'gpurun_out/prof'
RND = os.environ.get('RND', 'r05')
def trim(src, dst):
    rows = list(csv.reader(open(src)))
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:140]
            w.writerow(r)
for name in ('single', 'default', 'batch', 'batch_bf16', 'batch16', 'batch_k2b', 'quantized', 'funnel', 'multi', 'pattern_hamming', 'prefix_multi'):
    trim('%s/%s/p_kernel_stats.csv' % (out, name), '%s/%s_%s_kernel_stats.csv' % (out, RND, name))
def per_launch(path, kernel_substr, counter):
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(path))
            if kernel_substr in r['Kernel_Name'] and r['Counter_Name'] == counter]
    return sum(vals) / len(vals), len(vals)
def keep(path, dst, kernel_substr):
    rows = list(csv.DictReader(open(path)))
    with open(dst, 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows:
            if kernel_substr in r['Kernel_Name']:
                r['Kernel_Name'] = r['Kernel_Name'][:140]
                w.writerow(r)
keep(out + '/single_fetch/p_counter_collection.csv', out + '/%s_single_pmc_fetch.csv' % RND, 'scan_topk_kernel')
keep(out + '/single_write/p_counter_collection.csv', out + '/%s_single_pmc_write.csv' % RND, 'scan_topk_kernel')
keep(out + '/quantized_fetch/p_counter_collection.csv', out + '/%s_quantized_pmc_fetch.csv' % RND, 'hamming_dist_kernel')
keep(out + '/multi_fetch/p_counter_collection.csv', out + '/%s_multi_pmc_fetch.csv' % RND, 'scan_multi_kernel')
mf, mn = per_launch(out + '/multi_fetch/p_counter_collection.csv', 'scan_multi_kernel', 'FETCH_SIZE')
keep(out + '/prefix_multi_fetch/p_counter_collection.csv', out + '/%s_prefix_multi_pmc_fetch.csv' % RND, 'prefix_multi_kernel')
pf, pn = per_launch(out + '/prefix_multi_fetch/p_counter_collection.csv', 'prefix_multi_kernel', 'FETCH_SIZE')
print("K1p FETCH_SIZE KiB per launch (sample passes and sweeps mixed)", pf, "launches", pn)
print("K1m FETCH_SIZE KiB per sweep", mf, "x2 bytes", 2 * mf * 1024, "launches", mn)
fetch, n1 = per_launch(out + '/single_fetch/p_counter_collection.csv', 'scan_topk_kernel', 'FETCH_SIZE')
write, n2 = per_launch(out + '/single_write/p_counter_collection.csv', 'scan_topk_kernel', 'WRITE_SIZE')
stat = [r for r in csv.DictReader(open(out + '/single/p_kernel_stats.csv')) if 'scan_topk_kernel' in r['Name']][0]
json.dump({
    "round": int(RND[1:]),
    "command": "python3 bench.py --no-side  (N=10M, d=768, cosine, limit 10; PMC passes: --steps 20 --warmup 2 --no-cpu --no-side)",
    "rows": 10000000, "dim": 768, "kernel": stat['Name'],
    "rocprof_avg_ns": float(stat['AverageNs']), "rocprof_calls": int(stat['Calls']),
    "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 for 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; unit KiB",
    "hbm_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
}, open(out + '/pmc_latest.json', 'w'), indent=1)
hf, _ = per_launch(out + '/quantized_fetch/p_counter_collection.csv', 'hamming_dist_kernel', 'FETCH_SIZE')
print("hamming FETCH_SIZE KiB per launch", hf, "x2 bytes", 2 * hf * 1024)
ff, _ = per_launch(out + '/funnel_fetch/p_counter_collection.csv', 'cosine_scan_kernel', 'FETCH_SIZE')
bf, _ = per_launch(out + '/batch_fetch/p_counter_collection.csv', 'mfma_scores_kernel<8, false>', 'FETCH_SIZE')
# (the 256-column candidate pass of a 256-query batch: shadow_scores_kernel<MODE = 0, QTW = 8, ...> from the bf16
# shadow -- the default --, bf16_scores_kernel<DENSE = false, QT = 8> from the f32 rows with --shadow off)
K2S, K2B = 'shadow_scores_kernel<0, 8', 'bf16_scores_kernel<false, 8>'   # (K2s: <MODE = 0 the pass, QTW = 8, stages, DBG>)
s16, _ = per_launch(out + '/batch_bf16_fetch/p_counter_collection.csv', K2S, 'FETCH_SIZE')
keep(out + '/batch_bf16_fetch/p_counter_collection.csv', out + '/%s_batch_bf16_pmc_fetch.csv' % RND, K2S)
b16, _ = per_launch(out + '/batch_k2b_fetch/p_counter_collection.csv', K2B, 'FETCH_SIZE')
keep(out + '/batch_k2b_fetch/p_counter_collection.csv', out + '/%s_batch_k2b_pmc_fetch.csv' % RND, K2B)
print("shadow_scores_kernel FETCH_SIZE x2 bytes", 2 * s16 * 1024, "bf16_scores_kernel FETCH_SIZE x2 bytes", 2 * b16 * 1024)
keep(out + '/funnel_fetch/p_counter_collection.csv', out + '/%s_funnel_pmc_fetch.csv' % RND, 'cosine_scan_kernel')
keep(out + '/batch_fetch/p_counter_collection.csv', out + '/%s_batch_pmc_fetch.csv' % RND, 'mfma_scores_kernel<8, false>')
print("cosine_scan FETCH_SIZE x2 bytes", 2 * ff * 1024, "mfma_scores<8> FETCH_SIZE x2 bytes", 2 * bf * 1024)
# what the side legs of bench.py report as `traffic` (reads only: these kernels write a few KB of lists)
json.dump({k: {"rows": 10000000, "dim": 768, "hbm_bytes_per_launch": 2 * v * 1024,
               "source": "FETCH_SIZE x 2 (gfx950 correction as in pmc_latest.json), %s pass of tools/refresh_profiles.sh" % k}
           for k, v in (("hamming_dist_kernel", hf), ("cosine_scan_kernel", ff), ("mfma_scores_kernel", bf), ("scan_multi_kernel", mf),
                        ("bf16_scores_kernel", b16), ("shadow_scores_kernel", s16))},
          open(out + '/pmc_side.json', 'w'), indent=1)
import subprocess
with open('%s/%s_batch16_trace_excerpt.txt' % (out, RND), 'w') as f:
    f.write(subprocess.run(['python3', 'tools/trace_excerpt.py', glob.glob(out + '/batch16_trace/**/t_kernel_trace.csv', recursive=True)[0]],
                           capture_output=True, text=True).stdout)
for name in ('single', 'batch', 'batch_bf16', 'batch16', 'batch_k2b', 'quantized', 'funnel'):
    print(open('%s/%s.json' % (out, name)).read().strip())
    for r in csv.DictReader(open('%s/%s/p_kernel_stats.csv' % (out, name))):
        if 'vt::' in r['Name']:
            print('   ', r['Name'][:70], r['Calls'], r['AverageNs'])
PY
# counters of the two matrix-core passes (MFMA-busy and friends; VERDICT r3 missing #5): K2 on the FP32 pipe, K2s on the bf16 one
bash $R/tools/pmc_kernel.sh "mfma_scores_kernel<8, false>" $OUT/${RND}_k2_pmc.txt -- --mode batch --nominate f32 --steps 2 --warmup 1 --no-cpu
bash $R/tools/pmc_kernel.sh "shadow_scores_kernel<0, 8" $OUT/${RND}_k2s_pmc.txt -- --mode batch --nominate bf16 --steps 4 --warmup 1 --no-cpu
# probes without the profiler: K1p beside K1m and the single prefix scan; pattern funnels from the bit column and from the rows
cd /tmp
python3 $R/tools/prefix_multi_probe.py 2>/dev/null | grep "^{" > $OUT/${RND}_prefix_multi_probe.jsonl
ROWS=5000000 DIM=384 PREFIXES=128,384 python3 $R/tools/prefix_multi_probe.py 2>/dev/null | grep "^{" >> $OUT/${RND}_prefix_multi_probe.jsonl
python3 $R/tools/pattern_funnel_probe.py 2>/dev/null | grep "^{" > $OUT/${RND}_pattern_funnel_probe.jsonl
cd $R
# the 8-GPU operating points priced on one GPU (a shard of 10 M / 8 and of 40 M / 8 rows through worker + RCCL exchange + merge)
cd /tmp
python3 $R/bench.py --gpus 1 --exchange rccl --rows 1250000 --no-cpu > $OUT/${RND}_shard_10m_of_8.json 2> $OUT/shard_a.log
python3 $R/bench.py --gpus 1 --exchange rccl --metric l2 --rows 5000000 --no-cpu > $OUT/${RND}_shard_40m_of_8.json 2> $OUT/shard_b.log
python3 $R/bench.py --gpus 1 --exchange rccl --mode batch --metric l2 --rows 5000000 --no-cpu > $OUT/${RND}_shard_40m_of_8_batch.json 2> $OUT/shard_c.log
cd $R
tail -n 1 $OUT/${RND}_shard_10m_of_8.json $OUT/${RND}_shard_40m_of_8.json $OUT/${RND}_shard_40m_of_8_batch.json
