#!/bin/bash
OUT=gpurun_out/r03ao; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c3.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c3_kernel_stats.csv \; 2>/dev/null
grep "edge16t_kernel<false\|chain16_kernel" $OUT/c3_kernel_stats.csv | grep -v "4, 2>" | cut -c20-150
