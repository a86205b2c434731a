#!/bin/bash
OUT=gpurun_out/r03k; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_edge16.py tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -s -x --timeout 600 -p no:cacheprovider -k "round3 or edge16 or bf16 or c3 or determin or post or integration or team or fp16 or flat or graphcast" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -n 12 $OUT/pytest.log
timeout 400 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench_c3.log 2>&1; echo "rc=$?" >> $OUT/bench_c3.log; tail -n 2 $OUT/bench_c3.log | cut -c1-300
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_run.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c3_kernel_stats.csv \; 2>/dev/null
head -n 12 $OUT/c3_kernel_stats.csv | cut -c1-160
