#!/bin/bash
# round 3, call a: team-pipelined bf16 edge kernel - parity subset, phase clocks, c3 bench, kernel stats
OUT=gpurun_out/r03a; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_edge16.py tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -s -x --timeout 600 -p no:cacheprovider \
  -k "edge16 or bf16 or c3 or determin" > $OUT/pytest_bf16.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_bf16.log; tail -n 15 $OUT/pytest_bf16.log
timeout 200 python scripts/gpu_timeline16t.py 16 decoder > $OUT/timeline_decoder.log 2>&1; cat $OUT/timeline_decoder.log | tail -n 16
timeout 200 python scripts/gpu_timeline16t.py 16 processor > $OUT/timeline_processor.log 2>&1; cat $OUT/timeline_processor.log | tail -n 16
timeout 400 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench_c3.log 2>&1; echo "rc=$?" >> $OUT/bench_c3.log; tail -n 2 $OUT/bench_c3.log | cut -c1-1500
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_run.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c3_kernel_stats.csv \; 2>/dev/null
head -n 25 $OUT/c3_kernel_stats.csv | cut -c1-220
