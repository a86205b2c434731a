#!/bin/bash
OUT=gpurun_out/r03g; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_edge16.py -m gpu -q -s --timeout 600 -p no:cacheprovider > $OUT/pytest_r3.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_r3.log; tail -n 25 $OUT/pytest_r3.log
timeout 400 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench_c3.log 2>&1; echo "rc=$?" >> $OUT/bench_c3.log; tail -n 2 $OUT/bench_c3.log | cut -c1-300
