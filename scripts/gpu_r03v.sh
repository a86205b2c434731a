#!/bin/bash
# round 3, call v: gw_pack_many: tests, cold step, training step
OUT=gpurun_out/r03v; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "round3 or abi or parity or backward or narrow or alias" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -n 6 $OUT/pytest.log
timeout 300 python scripts/probes/cold_probe.py 2>&1 | grep cold | tee $OUT/cold.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > $OUT/train.log 2>&1; tail -n 1 $OUT/train.log | cut -c1-500
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/rocprof_train.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/train_kernel_stats.csv \; 2>/dev/null
head -n 24 $OUT/train_kernel_stats.csv | cut -c1-170
