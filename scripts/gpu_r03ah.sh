#!/bin/bash
OUT=gpurun_out/r03ah; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider -k "guards or parity or narrow or regional or alias" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -n 3 $OUT/pytest.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>&1 | tail -n 1 | cut -c1-330
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c2.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c2_kernel_stats.csv \; 2>/dev/null
head -n 9 $OUT/c2_kernel_stats.csv | cut -c1-150
