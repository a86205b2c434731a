#!/bin/bash
# round 3, call b: phase clocks of the team kernel under A/B switches (tuning build)
OUT=gpurun_out/r03b; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for which in decoder processor; do
  for cfg in "base" "GW_EDGE16_TUNE=1" "GW_EDGE16_SKIP=1" "GW_EDGE16_TUNE=2" "GW_EDGE16_TUNE=3 GW_EDGE16_SKIP=1"; do
    tag=$(echo "$cfg" | tr ' =' '__')
    if [ "$cfg" = "base" ]; then env_cmd=""; else env_cmd="env $cfg"; fi
    timeout 200 $env_cmd python scripts/gpu_timeline16t.py 16 $which > $OUT/tl_${which}_${tag}.log 2>&1
    echo "=== $which $cfg"; grep -v amdgpu.ids $OUT/tl_${which}_${tag}.log | tail -n 30
  done
done
