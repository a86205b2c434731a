#!/bin/bash
OUT=gpurun_out/r03w; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/probes/train_probe.py 2>&1 | grep -v amdgpu | tee $OUT/probe.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 2>&1 | tail -n 1 | cut -c1-400 | tee $OUT/train.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "round3 or narrow or regional or alias or parity" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -n 6 $OUT/pytest.log
