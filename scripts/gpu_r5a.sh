#!/bin/bash
# round 5, session A: split-mode tests + first timings, then the workgroup forms of the split kernels in a tuning build
TAG=${1:-r5a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
scripts/gpu_run.sh $TAG build "test:tests/test_gpu_split.py"
mv $OUT/pytest_gpu.log $OUT/pytest_split.log
scripts/gpu_run.sh $TAG "test:tests/test_gpu_round5.py"
mv $OUT/pytest_gpu.log $OUT/pytest_round5.log
scripts/gpu_run.sh $TAG "bench:--config+c2x3+--steps+20+--warmup+5+--no-extra+--no-cpu-baseline" "bench:--config+c3x3+--steps+10+--warmup+3+--no-extra+--no-cpu-baseline" "bench:--config+c2+--steps+20+--warmup+5+--no-extra+--no-cpu-baseline"
scripts/gpu_run.sh $TAG prof:c2x3
export GW_TUNING=1
python -c "import __graft_entry__ as g; g.build()" > $OUT/build_tuning.log 2>&1 || { echo TUNING BUILD FAILED; tail -20 $OUT/build_tuning.log; exit 0; }
for F in 41 42 81; do for FE in 41 42; do
  echo "== GW_X3_FORM=$F GW_X3_FORM_EDGE=$FE"
  GW_X3_FORM=$F GW_X3_FORM_EDGE=$FE timeout 300 python bench.py --config c2x3 --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>&1 | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('c2x3', d['value'], d['ms_per_step'], 'dec', r['launch_ms'], r['other_kernels_ms'])"
  GW_X3_FORM=$F GW_X3_FORM_EDGE=$FE timeout 300 python bench.py --config c3x3 --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>&1 | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('c3x3', d['value'], d['ms_per_step'], 'dec', r['launch_ms'], r['other_kernels_ms'])"
done; done > $OUT/forms.log 2>&1
cat $OUT/forms.log
