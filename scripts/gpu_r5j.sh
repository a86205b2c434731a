#!/bin/bash
# round 5, session J: split-operand TN GEMM (weight gradients of the mixed-precision training step)
TAG=${1:-r5j}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
scripts/gpu_run.sh $TAG build "test:tests/test_gpu_train_kernels.py" | tail -12
mv $OUT/pytest_gpu.log $OUT/pytest_train_kernels.log
scripts/gpu_run.sh $TAG "test:training" | tail -6
timeout 600 python bench.py --mode train --precision bf16x3 --steps 5 --warmup 2 2>&1 | tail -n 1 | cut -c1-330
timeout 600 python bench.py --mode train --steps 5 --warmup 2 2>&1 | tail -n 1 | cut -c1-330
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft -o gw -- python $GRAFT_REPO_ROOT/bench.py --mode train --precision bf16x3 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/rocprof_train_x3.log 2>&1
cd $GRAFT_REPO_ROOT; find /tmp/proft -name "*kernel_stats*.csv" -exec cp {} $OUT/train_x3_kernel_stats.csv \; ; head -n 12 $OUT/train_x3_kernel_stats.csv | cut -c1-170
