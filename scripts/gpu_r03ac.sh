#!/bin/bash
OUT=gpurun_out/r03ac; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_guards.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest.log | head -30
