#!/bin/bash
# One gpurun call: parity tests, smoke, bench, rocprofv3 kernel stats.  Everything is logged under gpurun_out/.
# usage: scripts/gpu_session.sh [tag]   (run from the repo root on the GPU box)
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{
  echo "== env"; date; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6; nproc; lscpu | grep -E "Model name|Socket|^CPU\(s\)"
  ls -la graph_weather_amd/csrc/
} > $OUT/env.log 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 --durations=15 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -n 40 $OUT/pytest_gpu.log
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -n 3 $OUT/smoke.log
timeout 900 python bench.py --steps ${BENCH_STEPS:-20} --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/bench.log; tail -n 3 $OUT/bench.log
if [ "${SKIP_PROF:-0}" != "1" ]; then
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra ${PROF_ARGS:-} > $GRAFT_REPO_ROOT/$OUT/rocprof_run.log 2>&1)
  echo "rocprof rc=$?" >> $OUT/rocprof_run.log
  find /tmp/prof -name "*stats*.csv" -exec cp {} $OUT/ \; 2>/dev/null
  find /tmp/prof -name "*kernel_trace.csv" -exec cp {} $OUT/ \; 2>/dev/null
  find /tmp/prof -name "*kernel_stats*" | head -1 | xargs -r head -n 20
fi
ls -la $OUT
