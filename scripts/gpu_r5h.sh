#!/bin/bash
# round 5, session H: DMA pieces over the first half of each chunk: tests, benches, waits
TAG=${1:-r5h}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
scripts/gpu_run.sh $TAG build "test:tests/test_gpu_split.py" | tail -3
mv $OUT/pytest_gpu.log $OUT/pytest_split.log
for C in c2x3 c3x3; do
  timeout 300 python bench.py --config $C --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>&1 | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$C', round(d['value'],1), round(d['ms_per_step'],3), 'dec', round(r['launch_ms'],3), r['other_kernels_ms'])"
done
export GW_TUNING=1
python -c "import __graft_entry__ as g; g.build()" > $OUT/build_tuning.log 2>&1 || { echo TUNING BUILD FAILED; tail -20 $OUT/build_tuning.log; exit 0; }
for FORMS in "41 41" "81 41"; do set -- $FORMS
  for C in c2x3 c3x3; do
  GW_X3_FORM=$1 GW_X3_FORM_EDGE=$2 timeout 300 python bench.py --config $C --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>&1 | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('forms $1 $2 $C', round(d['value'],1), round(d['ms_per_step'],3), 'dec', round(r['launch_ms'],3), r['other_kernels_ms'])"
  done
done
for W in decoder node; do
  GW_X3_FORM=41 GW_X3_FORM_EDGE=41 timeout 300 python scripts/gpu_timeline_x3.py 2 $W 2>&1 | grep -v "amdgpu.ids\|launch span"
done
