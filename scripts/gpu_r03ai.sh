#!/bin/bash
# round 3, call ai: anti-phase start of the two chain16 workgroups of a CU (tuning build: GW_STAGGER16 = sleep units)
OUT=gpurun_out/r03ai; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for st in 0 2 4 8 16; do
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && GW_STAGGER16=$st timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_s$st.log 2>&1)
  find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/stats_s$st.csv \; 2>/dev/null
  echo "stagger $st"; grep "chain16_kernel<" $OUT/stats_s$st.csv | grep "4, 1>" | cut -c40-135
done
