import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import perf_sweep as p
for nq in (8, 32, 64, 128):
    p.run_batch(3, 10_000_000, 768, nq, 10, False, steps=3)
