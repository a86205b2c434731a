#!/bin/bash
# round 3, call ad: per-kernel ablations of the chain16 launches inside the c3 forward (tuning build; results are wrong, timing only)
OUT=gpurun_out/r03ad; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for t in 0 1 2 3 4 7; do
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && GW_CHAIN16_TUNE=$t timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_t$t.log 2>&1)
  find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/stats_t$t.csv \; 2>/dev/null
  echo "tune $t" | tee -a $OUT/summary.log
  grep "chain16_kernel" $OUT/stats_t$t.csv | awk -F'","' '{printf "  %-95s %s us\n", substr($1,1,95), $4/1000}' | head -4 | tee -a $OUT/summary.log
done
