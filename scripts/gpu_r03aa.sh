#!/bin/bash
OUT=gpurun_out/r03aa; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/probes/graph_probe.py 2>&1 | grep -v amdgpu | tail -n 12 | tee $OUT/graph_probe.log
