#!/bin/bash
# round 5, session C: phase clocks of the split kernels (tuning build)
TAG=${1:-r5c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GW_TUNING=1
python -c "import __graft_entry__ as g; g.build()" > $OUT/build_tuning.log 2>&1 || { echo TUNING BUILD FAILED; tail -20 $OUT/build_tuning.log; exit 0; }
for F in 41 81; do
  for W in decoder processor node dechead nodeenc; do
    GW_X3_FORM=$F GW_X3_FORM_EDGE=$F timeout 300 python scripts/gpu_timeline_x3.py 2 $W 2>&1 | grep -v amdgpu.ids
  done
done > $OUT/timeline_b2.log 2>&1
GW_X3_FORM=41 GW_X3_FORM_EDGE=41 timeout 300 python scripts/gpu_timeline_x3.py 16 decoder 2>&1 | grep -v amdgpu.ids > $OUT/timeline_b16.log
GW_X3_FORM=41 GW_X3_FORM_EDGE=41 timeout 300 python scripts/gpu_timeline_x3.py 16 processor 2>&1 | grep -v amdgpu.ids >> $OUT/timeline_b16.log
cat $OUT/timeline_b2.log $OUT/timeline_b16.log
