#!/bin/bash
# One parameterised GPU runner (replaces the scripts/gpu_r0*.sh one-offs).  Usage, from the repo root on the GPU box:
#   scripts/gpu_run.sh TAG STEP [STEP ...]         output under gpurun_out/TAG/
# Steps:
#   build                  __graft_entry__.build()
#   test[:EXPR]            pytest -m gpu (EXPR = -k expression or a test file path)
#   smoke                  __graft_entry__.smoke()
#   bench[:ARGS]           python bench.py ARGS (default: the driver's command, --steps 20 --warmup 5); ARGS with '+' for spaces
#   prof:CONFIG            rocprofv3 --kernel-trace --stats of bench.py --config CONFIG --steps 5 (kernel_stats csv copied)
#   proftrain[:PRECISION]  rocprofv3 --kernel-trace --stats of the training step (bench.py --mode train --steps 3; fp32 or bf16x3)
#   pmc:c3 | pmc:c2        the multi-pass PMC scripts (scripts/gpu_pmc_c3.sh, scripts/gpu_pmc.sh): summary json under gpurun_out/TAG
#   py:SCRIPT[:ARGS]       python SCRIPT ARGS (probes under scripts/probes)
# Tuning builds: GW_TUNING=1 in the environment of the call rebuilds the library with the A/B knobs.
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
for STEP in "$@"; do
  KIND=${STEP%%:*}; ARG=""; [ "$STEP" != "$KIND" ] && ARG=${STEP#*:}
  case $KIND in
    build)
      python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; } ;;
    test)
      SEL=(); if [ -n "$ARG" ]; then if [ -e "$ARG" ]; then SEL=("$ARG"); else SEL=(tests -k "$ARG"); fi; else SEL=(tests); fi
      timeout 1700 python -X faulthandler -m pytest "${SEL[@]}" -m gpu -q -s --timeout 900 --durations=8 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
      echo "pytest rc=$?" >> $OUT/pytest_gpu.log
      grep -E "passed|failed|error|rc=|FAILED|Error" $OUT/pytest_gpu.log | tail -n 15
      grep -E "^\[" $OUT/pytest_gpu.log | cut -c1-220 | tail -n 40 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -n 2 $OUT/smoke.log ;;
    bench)
      A=${ARG:-"--steps+20+--warmup+5"}; N=$(echo "$A" | tr -c 'a-zA-Z0-9' '_' | cut -c1-40)
      timeout 1200 python bench.py ${A//+/ } > $OUT/bench_$N.log 2>&1; echo "bench rc=$?" >> $OUT/bench_$N.log; tail -n 2 $OUT/bench_$N.log | cut -c1-3500 ;;
    prof)
      rm -rf /tmp/prof && mkdir -p /tmp/prof
      (cd /tmp && GW_AUTO_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $R/bench.py --config $ARG --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $R/$OUT/rocprof_$ARG.log 2>&1)  # (eager launches: the same kernels the HIP graph replays)
      find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/${ARG}_kernel_stats.csv \; 2>/dev/null
      tail -n 1 $OUT/rocprof_$ARG.log | cut -c1-600; head -n 16 $OUT/${ARG}_kernel_stats.csv | cut -c1-200 ;;
    proftrain)
      P=${ARG:-fp32}; rm -rf /tmp/prof && mkdir -p /tmp/prof
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $R/bench.py --mode train --precision $P --steps 3 --warmup 2 > $R/$OUT/rocprof_train_$P.log 2>&1)
      find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/train_${P}_kernel_stats.csv \; 2>/dev/null
      tail -n 1 $OUT/rocprof_train_$P.log | cut -c1-400; head -n 12 $OUT/train_${P}_kernel_stats.csv | cut -c1-200 ;;
    pmc)  # pmc:c3 = HBM traffic / MFMA activity of the bf16 edge kernels per position in the forward; pmc:c2 = the fp32 decoder edge kernel
      if [ "$ARG" = "c3" ]; then bash scripts/gpu_pmc_c3.sh $TAG; else bash scripts/gpu_pmc.sh $TAG; fi ;;
    py)
      S=${ARG%%:*}; A=""; [ "$ARG" != "$S" ] && A=${ARG#*:}; N=$(basename $S .py)
      timeout 900 python $S ${A//+/ } > $OUT/$N.log 2>&1; echo "rc=$?" >> $OUT/$N.log; tail -n 40 $OUT/$N.log | cut -c1-300 ;;
    *) echo "unknown step $STEP" ;;
  esac
done
