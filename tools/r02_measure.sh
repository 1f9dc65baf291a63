#!/bin/bash
# Round-2 measurements quoted in DESIGN.md section 5 (one MI355X box; run through gpurun).
# Output: gpurun_out/r02/*.json(l)
set -u
mkdir -p gpurun_out/r02
O=gpurun_out/r02
for ord in pair avx seq sse2; do
  timeout 300 python bench.py --no-side --no-cpu --reduce-order $ord 2>/dev/null | tail -1 > $O/order_$ord.json
done
timeout 300 python bench.py --gpus 2 --devices 0,0 --no-cpu 2>/dev/null | tail -1 > $O/two_shards_one_gpu.json
timeout 300 python bench.py --gpus 1 --exchange rccl --no-cpu 2>/dev/null | tail -1 > $O/one_shard_rccl.json
timeout 300 python bench.py --gpus 1 --exchange host --no-cpu 2>/dev/null | tail -1 > $O/one_shard_host.json
timeout 600 python -m pytest tests/test_gpu_perf.py -m gpu_perf -q -s 2>&1 | tail -8 > $O/gpu_perf.log
ROWS=10000000 NQS=1,8,16,64 METRICS=5,2 timeout 300 python tools/multi_probe.py 2>/dev/null | grep "^{" > $O/multi_probe_10m.jsonl
timeout 300 python tools/cliff_probe.py 2>/dev/null | tail -40 > $O/cliff_probe.log
