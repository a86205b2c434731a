#!/bin/bash
# round 5, session F: fragment ring depth 3 vs 5 (tuning build), PMC of the split kernels at c2x3
TAG=${1:-r5f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GW_TUNING=1
python -c "import __graft_entry__ as g; g.build()" > $OUT/build_tuning.log 2>&1 || { echo TUNING BUILD FAILED; tail -20 $OUT/build_tuning.log; exit 0; }
GW_X3_RING=5 timeout 900 python -m pytest tests/test_gpu_split.py -m gpu -q -x -p no:cacheprovider -k "not c5 and not c2_and_c3" > $OUT/pytest_split_ring5.log 2>&1; tail -n 2 $OUT/pytest_split_ring5.log
for RING in 3 5; do for FORMS in "41 41" "81 81"; do set -- $FORMS
  echo "== GW_X3_RING=$RING GW_X3_FORM=$1 GW_X3_FORM_EDGE=$2"
  for C in c2x3 c3x3; do
  GW_X3_RING=$RING GW_X3_FORM=$1 GW_X3_FORM_EDGE=$2 timeout 300 python bench.py --config $C --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>&1 | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$C', round(d['value'],1), round(d['ms_per_step'],3), 'dec', round(r['launch_ms'],3), r['other_kernels_ms'])"
  done
  for W in decoder node; do
    GW_X3_RING=$RING GW_X3_FORM=$1 GW_X3_FORM_EDGE=$2 timeout 300 python scripts/gpu_timeline_x3.py 2 $W 2>&1 | grep -v "amdgpu.ids\|launch span"
  done
done; done > $OUT/ring.log 2>&1
cat $OUT/ring.log
GW_X3_FORM=41 GW_X3_FORM_EDGE=41 bash scripts/gpu_pmc_cfg.sh $TAG c2x3 > $OUT/pmc_c2x3_summary.log 2>&1
cat $OUT/pmc_c2x3_summary.log | cut -c1-700
