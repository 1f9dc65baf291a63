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
export RND=${RND:-r06}
stats single --no-side          # the headline alone: the scan kernel's average is the bench line's avg_launch_ms
stats default                   # no flags (1 000 timed steps): headline + side legs (config 2 shares the scan kernel: 1 100 launches at N=1M)
stats driver_cmd --gpus 1 --steps 20 --warmup 5   # LITERALLY the driver's command under the profiler (VERDICT r5 #4a): the line's avg_launch_ms beside rocprofv3's own average
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
# K1m: 8 queries per sweep (manhattan, N=10M, d=256 -- since r05 rows of 512 floats and more of a corpus this size go to K1p,
# and the switch that forced K1m there left in r06: the profile is of a shape the product sends to K1m); the program
# after `--` is python3 itself
export ROWS=10000000 DIM=256 NQS=8 METRICS=5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/multi -o p -- python3 $R/tools/multi_probe.py > $OUT/multi.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/multi_fetch -o p -- python3 $R/tools/multi_probe.py > $OUT/multi_fetch.log 2>&1
unset ROWS DIM NQS METRICS
# K1p: eight L2 funnel searches per sweep of the 128-float prefixes (N=10M, d=768) -- stats pass, FETCH pass
export METRICS=0 PREFIXES=128 K1M=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefix_multi -o p -- python3 $R/tools/prefix_multi_probe.py > $OUT/prefix_multi.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prefix_multi_fetch -o p -- python3 $R/tools/prefix_multi_probe.py > $OUT/prefix_multi_fetch.log 2>&1
unset METRICS PREFIXES K1M
cd $R
python3 tools/refresh_post.py
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
python3 $R/bench.py --gpus 1 --exchange rccl --mode batch --metric l2 --rows 5000000 --steps 60 --warmup 5 --no-cpu > $OUT/${RND}_shard_40m_of_8_batch.json 2> $OUT/shard_c.log
python3 $R/bench.py --gpus 1 --exchange rccl --mode batch --metric l2 --rows 5000000 --batch 4096 --steps 6 --warmup 1 --no-cpu > $OUT/${RND}_shard_40m_of_8_batch16.json 2> $OUT/shard_d.log
cd $R
tail -n 1 $OUT/${RND}_shard_10m_of_8.json $OUT/${RND}_shard_40m_of_8.json $OUT/${RND}_shard_40m_of_8_batch.json $OUT/${RND}_shard_40m_of_8_batch16.json
