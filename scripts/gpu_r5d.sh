#!/bin/bash
# round 5, session D: ablations of the split pass (tuning build; wrong results, timing only)
TAG=${1:-r5d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GW_TUNING=1
python -c "import __graft_entry__ as g; g.build()" > $OUT/build_tuning.log 2>&1 || { echo TUNING BUILD FAILED; tail -20 $OUT/build_tuning.log; exit 0; }
for F in 41 81; do
for ABL in 0 1 2 8 4 9 10 11 3; do
  for W in decoder node; do
    echo "## form $F ABL=$ABL (1 no DMA, 2 no fragment reads, 4 no barriers, 8 no MFMAs)"
    GW_X3_ABL=$ABL GW_X3_FORM=$F GW_X3_FORM_EDGE=$F timeout 300 python scripts/gpu_timeline_x3.py 2 $W 2>&1 | grep -E "batch|\[2\]|\[3\]|\[6\]"
  done
done; done > $OUT/ablation.log 2>&1
cat $OUT/ablation.log
