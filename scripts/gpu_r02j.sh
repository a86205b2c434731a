#!/bin/bash
# Evidence refresh after the gather fusion: rocprofv3 kernel stats (c3, c2), phase clocks of the resident kernels, PMC (c3).
OUT=gpurun_out/r02j; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
for cfg in c3 c2; do
  rm -rf /tmp/prof_$cfg && mkdir -p /tmp/prof_$cfg
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o gw -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_$cfg.log 2>&1)
  find /tmp/prof_$cfg -name "*kernel_stats*" -exec cp {} $OUT/${cfg}_kernel_stats.csv \; 2>/dev/null
  echo "== $cfg"; head -n 14 $OUT/${cfg}_kernel_stats.csv | cut -c1-170
  tail -n 1 $OUT/rocprof_$cfg.log | cut -c1-200
done
timeout 300 python scripts/gpu_timeline16.py 16 decoder > $OUT/timeline_decoder.log 2>&1; tail -n 25 $OUT/timeline_decoder.log
timeout 300 python scripts/gpu_timeline16.py 16 processor > $OUT/timeline_processor.log 2>&1; tail -n 25 $OUT/timeline_processor.log
bash scripts/gpu_pmc_c3.sh r02j/pmc_c3 > $OUT/pmc_c3.log 2>&1; tail -n 40 $OUT/pmc_c3.log | cut -c1-250
