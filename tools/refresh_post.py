#!/usr/bin/env python3
"""The post-processing step of tools/refresh_profiles.sh (kernel-stats files cut to 140-character names, per-launch
FETCH_SIZE / WRITE_SIZE figures, pmc_latest.json / pmc_side.json, the 16 x 256 trace excerpt): reads gpurun_out/prof/,
round from $RND.  A file of its own so that it can be re-run on the merged data without a GPU."""
import csv, glob, json, os
out = 'gpurun_out/prof'
RND = os.environ.get('RND', 'r06')
def trim(src, dst):
    rows = list(csv.reader(open(src)))
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:140]
            w.writerow(r)
for name in ('single', 'default', 'driver_cmd', 'batch', 'batch_bf16', 'batch16', 'batch_k2b', 'quantized', 'funnel', 'multi', 'pattern_hamming', 'prefix_multi'):
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
           for k, v in (("hamming_dist_kernel", hf), ("cosine_scan_kernel", ff), ("mfma_scores_kernel", bf),
                        ("bf16_scores_kernel", b16), ("shadow_scores_kernel", s16))},
          open(out + '/pmc_side.json', 'w'), indent=1)
side = json.load(open(out + '/pmc_side.json'))
# (K1m is profiled at d = 256 since r06: tools/refresh_profiles.sh says why)
side["scan_multi_kernel"] = {"rows": 10000000, "dim": 256, "hbm_bytes_per_launch": 2 * mf * 1024,
                             "source": "FETCH_SIZE x 2 (gfx950 correction as in pmc_latest.json), scan_multi_kernel pass of tools/refresh_profiles.sh"}
json.dump(side, open(out + '/pmc_side.json', 'w'), indent=1)
import subprocess
with open('%s/%s_batch16_trace_excerpt.txt' % (out, RND), 'w') as f:
    f.write(subprocess.run(['python3', 'tools/trace_excerpt.py', glob.glob(out + '/batch16_trace/**/t_kernel_trace.csv', recursive=True)[0]],
                           capture_output=True, text=True).stdout)
import shutil
shutil.copyfile('%s/default.json' % out, '%s/%s_bench_default_under_rocprof.json' % (out, RND))
for name in ('single', 'batch', 'batch_bf16', 'batch16', 'batch_k2b', 'quantized', 'funnel'):
    # (the JSON line of every profiled leg under the name profiles/ keeps it by: <round>_bench_<leg>.json)
    shutil.copyfile('%s/%s.json' % (out, name), '%s/%s_bench_%s.json' % (out, RND, name))
# (the driver's literal command: <round>_bench_driver_cmd.json is its UN-profiled line, written by hand from a plain run)
shutil.copyfile('%s/driver_cmd.json' % out, '%s/%s_bench_driver_cmd_under_rocprof.json' % (out, RND))
with open('%s/%s_driver_cmd_trace_summary.txt' % (out, RND), 'w') as f:
    f.write(subprocess.run(['python3', 'tools/driver_cmd_trace_summary.py'], capture_output=True, text=True).stdout)
    print(open('%s/%s.json' % (out, name)).read().strip())
    for r in csv.DictReader(open('%s/%s/p_kernel_stats.csv' % (out, name))):
        if 'vt::' in r['Name']:
            print('   ', r['Name'][:70], r['Calls'], r['AverageNs'])
