#!/bin/bash
# round 3, call s: chain16 4 x 1 with batch-innermost tiles: c3 bench, kernel stats, bf16 tests
OUT=gpurun_out/r03s; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 120 python scripts/probes/chain16_probe.py 2>&1 | grep rows | tee -a $OUT/probe.log
timeout 300 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench_c3.log 2>&1
tail -n 1 $OUT/bench_c3.log | cut -c1-400
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c3.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c3_kernel_stats.csv \; 2>/dev/null
head -n 9 $OUT/c3_kernel_stats.csv | cut -c1-170
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "bf16 or round3 or c3 or narrow or guards" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -n 6 $OUT/pytest.log
