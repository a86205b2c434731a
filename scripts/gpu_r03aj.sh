#!/bin/bash
# round 3, call aj: persistent 64-column chain16 (tuning build: GW_CHAIN16_WGS = workgroups; 1000000 = one per tile)
OUT=gpurun_out/r03aj; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for w in 1000000 512 1024; do
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && GW_CHAIN16_WGS=$w timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_w$w.log 2>&1)
  find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/stats_w$w.csv \; 2>/dev/null
  echo "workgroups $w"; grep "chain16_kernel<" $OUT/stats_w$w.csv | grep "4, 1>" | cut -c40-135
done
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "guards or c3 or bf16 or round3 or narrow" 2>&1 | tail -n 3
