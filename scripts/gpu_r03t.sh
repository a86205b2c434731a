#!/bin/bash
# round 3, call t: full GPU suite, smoke, default bench line (driver contract), kernel stats of c2 and c3, c3 counters
OUT=gpurun_out/r03t; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ date; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6; nproc; lscpu | grep -E "Model name|Socket|^CPU\(s\)"; } > $OUT/env.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 --durations=12 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -n 5 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -n 2 $OUT/smoke.log
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c2.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c2_kernel_stats.csv \; 2>/dev/null
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c3.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c3_kernel_stats.csv \; 2>/dev/null
head -n 9 $OUT/c3_kernel_stats.csv | cut -c1-150
bash scripts/gpu_pmc_c3.sh r03t_pmc > $OUT/pmc.log 2>&1
mkdir -p profiles_tmp; cp gpurun_out/r03t_pmc/pmc_c3.json profiles/r03_pmc_c3.json 2>/dev/null
timeout 1200 python bench.py > $OUT/bench_default.log 2>&1; echo "bench rc=$?" >> $OUT/bench_default.log; tail -n 2 $OUT/bench_default.log | cut -c1-300
