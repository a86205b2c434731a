#!/bin/bash
# round 3, call y: layer-1 kernel with two groups per wave (tuning build: GW_L1_GD = row pieces in flight)
OUT=gpurun_out/r03y; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for gd in 8 4; do
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && GW_L1_GD=$gd timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c3_gd$gd.log 2>&1)
  find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c3_kernel_stats_gd$gd.csv \; 2>/dev/null
  grep "l1_kernel\|edge16t_kernel<false" $OUT/c3_kernel_stats_gd$gd.csv | cut -c1-170
done
timeout 300 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>&1 | tail -n 1 | cut -c1-330
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "bf16 or round3 or c3 or edge16" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -n 4 $OUT/pytest.log
