for d in 0 14 15 30 31; do
  echo "debug=$d"; VT_BATCH_DEBUG=$d timeout 300 python bench.py --mode batch --rows 10000000 --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['achieved'], j['roofline']['avg_launch_ms'])"
done
