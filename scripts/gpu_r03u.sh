#!/bin/bash
# round 3, call u: kernel stats of cold forwards (weights just changed)
OUT=gpurun_out/r03u; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/scripts/probes/cold_probe.py > $GRAFT_REPO_ROOT/$OUT/cold.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/cold_kernel_stats.csv \; 2>/dev/null
tail -n 2 $OUT/cold.log; head -n 30 $OUT/cold_kernel_stats.csv | cut -c1-200
