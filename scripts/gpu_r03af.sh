#!/bin/bash
OUT=gpurun_out/r03af; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for t in 0 7; do
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && GW_CHAIN16_TUNE=$t timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_t$t.log 2>&1)
  find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/stats_t$t.csv \; 2>/dev/null
  echo "tune $t"; grep "chain16_kernel<" $OUT/stats_t$t.csv | grep "4, 1>" | cut -c40-150
done
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "guards or c3 or bf16 or round3" 2>&1 | tail -n 2
