#!/bin/bash
# full GPU suite + smoke + default bench (what the driver runs at round end)
TAG=${1:-full}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 1700 python -X faulthandler -m pytest tests -m gpu -q -s --timeout 900 --durations=12 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|error|rc=|FAILED|Error" $OUT/pytest_gpu.log | tail -n 15
grep -E "^\[|^[0-9.]+s (call|setup)" $OUT/pytest_gpu.log | cut -c1-220 | tail -n 60
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -n 2 $OUT/smoke.log
if [ "${SKIP_BENCH:-0}" != "1" ]; then
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/bench.log; tail -n 2 $OUT/bench.log | cut -c1-3000
fi
