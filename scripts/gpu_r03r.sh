#!/bin/bash
# round 3, call r: chain16 ablations at 8 x 1; 4 x 1 with two workgroups per CU (tuning build)
OUT=gpurun_out/r03r; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for nw in 8 41; do
  for t in 0 1 2 3 4 7; do
  GW_CHAIN16_NW=$nw GW_CHAIN16_TUNE=$t timeout 120 python scripts/probes/chain16_probe.py 2>&1 | grep rows | sed "s/$/ NW=$nw/" | tee -a $OUT/probe.log
  done
done
for nw in 41; do
  GW_CHAIN16_NW=$nw timeout 300 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench_c3_nw$nw.log 2>&1
  tail -n 1 $OUT/bench_c3_nw$nw.log | cut -c1-400
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && GW_CHAIN16_NW=$nw timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c3_nw$nw.log 2>&1)
  find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c3_kernel_stats_nw$nw.csv \; 2>/dev/null
  head -n 9 $OUT/c3_kernel_stats_nw$nw.csv | cut -c1-170
done
