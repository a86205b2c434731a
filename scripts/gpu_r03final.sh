#!/bin/bash
# round 3, final: full GPU suite, smoke, kernel stats of c2 / c3 / the training step, c3 counters, default bench line (driver contract)
OUT=gpurun_out/r03final; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ date; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6; nproc; lscpu | grep -E "Model name|Socket|^CPU\(s\)"; } > $OUT/env.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 --durations=12 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -n 5 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -n 2 $OUT/smoke.log
for cfg in c2 c3; do
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_$cfg.log 2>&1)
  find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/${cfg}_kernel_stats.csv \; 2>/dev/null
done
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/rocprof_train.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/train_kernel_stats.csv \; 2>/dev/null
head -n 9 $OUT/c3_kernel_stats.csv | cut -c1-150
bash scripts/gpu_pmc_c3.sh r03final_pmc > $OUT/pmc.log 2>&1
cp gpurun_out/r03final_pmc/pmc_c3.json profiles/r03_pmc_c3.json 2>/dev/null
timeout 300 python scripts/probes/train_probe.py 2>&1 | grep -v amdgpu > $OUT/train_probe.log
timeout 300 python scripts/probes/cold_probe.py 2>&1 | grep cold > $OUT/cold_probe.log
timeout 1200 python bench.py > $OUT/bench_default.log 2>&1; echo "bench rc=$?" >> $OUT/bench_default.log; tail -n 2 $OUT/bench_default.log | cut -c1-300
