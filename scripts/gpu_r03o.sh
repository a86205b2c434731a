#!/bin/bash
OUT=gpurun_out/r03o; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for t in 0 1 2 3 4 5 6 7; do GW_CHAIN16_TUNE=$t timeout 120 python scripts/probes/chain16_probe.py 2>&1 | grep rows | tee -a $OUT/probe.log; done
