#!/bin/bash
# round 5, session G: where the chunk hand-overs of the split pass wait (tuning build)
TAG=${1:-r5g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GW_TUNING=1
python -c "import __graft_entry__ as g; g.build()" > $OUT/build_tuning.log 2>&1 || { echo TUNING BUILD FAILED; tail -20 $OUT/build_tuning.log; exit 0; }
for FORMS in "41 41" "81 81"; do set -- $FORMS
  for W in decoder node processor; do
    GW_X3_FORM=$1 GW_X3_FORM_EDGE=$2 timeout 300 python scripts/gpu_timeline_x3.py 2 $W 2>&1 | grep -v "amdgpu.ids\|launch span"
  done
done > $OUT/waits.log 2>&1
cat $OUT/waits.log
