#!/bin/bash
OUT=gpurun_out/r03am; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider -k "backward or train or gradients or regional" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -n 3 $OUT/pytest.log
timeout 300 python scripts/probes/train_probe.py 2>&1 | grep -v amdgpu | tail -n 3
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/rocprof_train.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/train_kernel_stats.csv \; 2>/dev/null
grep "ln_bwd\|relu_bwd" $OUT/train_kernel_stats.csv | cut -c1-60,140-230
