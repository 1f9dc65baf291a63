#!/bin/bash
# K2 timing experiments: VT_BATCH_DEBUG bits -- 2: no per-chunk barrier, 4: no per-chunk waits,
# 8: no candidate append (results are invalid with any bit set; only the kernel time is meaningful).
# Needs the experiments build (`make experiments`: vettore_amd/lib/experiments/libvettore_hip.so); the product library
# has neither the switch nor the code behind it.
export VETTORE_HIP_LIB=${VETTORE_HIP_LIB:-$(dirname $0)/../vettore_amd/lib/experiments/libvettore_hip.so}
ROWS=${ROWS:-10000000}
for d in ${DEBUGS:-0 8 6 14}; do
  echo "debug=$d"; VT_BATCH_DEBUG=$d timeout 300 python bench.py --mode batch --rows $ROWS --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['achieved'], j['roofline']['avg_launch_ms'])"
done
