#!/bin/bash
# round 3, call e: MFMA phases of the team kernel in isolation (tuning build; results wrong under these switches)
OUT=gpurun_out/r03e; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for which in decoder processor; do
  for cfg in "GW_EDGE16_TUNE=4" "GW_EDGE16_TUNE=24" "GW_EDGE16_TUNE=28"; do
    tag=$(echo "$cfg" | tr ' =' '__')
    timeout 200 env $cfg python scripts/gpu_timeline16t.py 16 $which > $OUT/tl_${which}_${tag}.log 2>&1
    echo "=== $which $cfg"; grep -v amdgpu.ids $OUT/tl_${which}_${tag}.log | grep -E "team|mid group 0|out group 0|segment sums|LN:|step total|wait at"
  done
done
