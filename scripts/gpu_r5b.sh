#!/bin/bash
# round 5, session B: split-mode tests on the new forms, then form A/B in a tuning build
TAG=${1:-r5b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
scripts/gpu_run.sh $TAG build "test:tests/test_gpu_split.py"
mv $OUT/pytest_gpu.log $OUT/pytest_split.log
scripts/gpu_run.sh $TAG "bench:--config+c2x3+--steps+20+--warmup+5+--no-extra+--no-cpu-baseline" "bench:--config+c3x3+--steps+10+--warmup+3+--no-extra+--no-cpu-baseline"
scripts/gpu_run.sh $TAG prof:c2x3
export GW_TUNING=1
python -c "import __graft_entry__ as g; g.build()" > $OUT/build_tuning.log 2>&1 || { echo TUNING BUILD FAILED; tail -20 $OUT/build_tuning.log; exit 0; }
# the split tests again with every launch forced onto the 8-wave form (small launches included)
GW_X3_FORM=81 GW_X3_FORM_EDGE=81 timeout 900 python -m pytest tests/test_gpu_split.py -m gpu -q -x -p no:cacheprovider -k "not c5 and not c2_and_c3" > $OUT/pytest_split_form81.log 2>&1; tail -n 3 $OUT/pytest_split_form81.log
for FORMS in "0 0" "41 41" "81 81" "81 41" "41 81"; do set -- $FORMS
  echo "== GW_X3_FORM=$1 GW_X3_FORM_EDGE=$2"
  for C in c2x3 c3x3; do
  GW_X3_FORM=$1 GW_X3_FORM_EDGE=$2 timeout 300 python bench.py --config $C --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>&1 | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$C', round(d['value'],1), round(d['ms_per_step'],3), 'dec', round(r['launch_ms'],3), r['other_kernels_ms'])"
  done
done > $OUT/forms.log 2>&1
cat $OUT/forms.log
