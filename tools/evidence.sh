#!/bin/bash
# The round's run records that the driver does not produce itself (VERDICT r4 #3c): the `-m gpu_perf` guards and the
# soaks on the build that is in the tree, kept under profiles/<round>/.
#   gpurun --timeout 2700 -- 'RND=r06 bash tools/evidence.sh'      (then: cp -r gpurun_out/evidence/r06/* profiles/r06/)
# SOAK_SCALE (default 1.0) scales every soak's seconds; PERF=0 / SOAKS=0 skip a part; SUITE=<n> adds n runs of the GPU suite.
R=${GRAFT_REPO_ROOT:-$(pwd)}
RND=${RND:-r06}
OUT=$R/gpurun_out/evidence/$RND
mkdir -p $OUT
cd $R
S=${SOAK_SCALE:-1.0}
secs() { python3 -c "print(max(20, int($1 * $S)))"; }
# SUITE=<n>: the whole `-m gpu` suite n times on THIS lease (VERDICT r5 #1d: five runs over at least three leases); every
# run's verdict line goes to gpu_suite_runs.<lease>.log with the GPU's unique id -- the logs of the round's leases are
# concatenated into profiles/<round>/gpu_suite_runs.log.
if [ "${SUITE:-0}" != "0" ]; then
  LEASE=$(date -u +%Y%m%dT%H%M%SZ)
  GPUID=$(rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id" | grep -v "====" | head -1 | sed 's/.*: *//')
  for i in $(seq 1 $SUITE); do
    t0=$(date +%s)
    timeout 1700 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/suite_${LEASE}_$i.full 2>&1
    rc=$?
    echo "$(date -u +%Y-%m-%dT%H:%MZ) lease $LEASE gpu ${GPUID:-?} run $i: rc $rc, $(grep -E "[0-9]+ (passed|failed)" $OUT/suite_${LEASE}_$i.full | tail -1) [wall $(( $(date +%s) - t0 )) s, pytest -m gpu -x on $(git -C $R rev-parse --short HEAD 2>/dev/null || echo 'the tree as sent')]" | tee -a $OUT/gpu_suite_runs.$LEASE.log
    [ $rc -ne 0 ] && tail -60 $OUT/suite_${LEASE}_$i.full
  done
fi
if [ "${PERF:-1}" != "0" ]; then
  # (-s: the guards print what they measured)
  timeout 1500 python3 -m pytest tests -m gpu_perf -s -q > $OUT/gpu_perf.log 2>&1
  tail -3 $OUT/gpu_perf.log
fi
if [ "${SOAKS:-1}" != "0" ]; then
  {
    echo "# $(date -u +%Y-%m-%dT%H:%MZ) soaks on $(git -C $R rev-parse --short HEAD 2>/dev/null || echo 'the tree as sent'), SOAK_SCALE=$S"
    echo "# soak_groups (grouped paths: funnel batches under nine metrics, plain batches as K1p sweeps)"
    SECONDS=$(secs 240) python3 tools/soak_groups.py 2>&1 | tail -1
    echo "# soak_groups on corpora of 150 000-400 000 rows (the groups' thresholds from tile maxima, as on a 10 M-row corpus)"
    ROWS_MIN=150000 ROWS_MAX=400000 SECONDS=$(secs 150) python3 tools/soak_groups.py 2>&1 | tail -1
    echo "# soak_all (every search entry point, plain handle)"
    SECONDS=$(secs 200) python3 tools/soak_all.py 2>&1 | tail -1
    echo "# soak_all SHARDS=3"
    SECONDS=$(secs 150) SHARDS=3 python3 tools/soak_all.py 2>&1 | tail -1
    echo "# soak_all METRICS=1,4,6,7,8"
    SECONDS=$(secs 150) METRICS=1,4,6,7,8 python3 tools/soak_all.py 2>&1 | tail -1
    echo "# soak_mutations"
    SECONDS=$(secs 120) python3 tools/soak_mutations.py 2>&1 | tail -1
    echo "# soak_pattern"
    SECONDS=$(secs 90) python3 tools/soak_pattern.py 2>&1 | tail -1
    echo "# soak_upsert_search"
    SECONDS=$(secs 90) python3 tools/soak_upsert_search.py 2>&1 | tail -1
    echo "# soak_concurrent (8 readers + a writer), then SHARDS=3"
    SECONDS=$(secs 150) python3 tools/soak_concurrent.py 2>&1 | tail -1
    SECONDS=$(secs 120) SHARDS=3 python3 tools/soak_concurrent.py 2>&1 | tail -1
    echo "# soak_batch_pipeline (r05: batch calls of 2..5 groups of 256 on two contexts, every list against the oracle)"
    SECONDS=$(secs 150) python3 tools/soak_batch_pipeline.py 2>&1 | tail -1
    echo "# soak_batch_pipeline SHARDS=3 (the same calls on a sharded handle: the calling thread merges under the later groups)"
    SHARDS=3 SECONDS=$(secs 90) python3 tools/soak_batch_pipeline.py 2>&1 | tail -1
  } > $OUT/soaks.log 2>&1
  cat $OUT/soaks.log
fi
