#!/bin/bash
OUT=gpurun_out/r03al; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for t in 0 32 0 32; do
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && GW_EDGE16_TUNE=$t timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c3_$t.log 2>&1)
  find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c3_kernel_stats_$t.csv \; 2>/dev/null
  echo "tune $t"; grep "edge16t_kernel<false" $OUT/c3_kernel_stats_$t.csv | cut -c1-150
done
