#!/bin/bash
OUT=gpurun_out/r03ab; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "1024 256" "512 256" "512 512" "256 512" "2048 256" "768 1024"; do
  set -- $cfg
  GW_TN_TARGET=$1 GW_TN_MIN=$2 timeout 300 python bench.py --mode train --steps 5 --warmup 2 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('target $1 min $2:', round(d['ms_per_step'],2), 'ms')" | tee -a $OUT/tn.log
done
