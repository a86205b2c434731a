#!/bin/bash
# round 3, call f: team kernel v3 (one filler per MFMA) - parity subset, phase clocks, c3 bench
OUT=gpurun_out/r03f; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_edge16.py tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -s -x --timeout 600 -p no:cacheprovider \
  -k "edge16 or bf16 or c3 or determin" > $OUT/pytest_bf16.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_bf16.log; tail -n 6 $OUT/pytest_bf16.log
for which in decoder processor; do
  for cfg in "base" "GW_EDGE16_TUNE=28"; do
    tag=$(echo "$cfg" | tr ' =' '__')
    if [ "$cfg" = "base" ]; then env_cmd=""; else env_cmd="env $cfg"; fi
    timeout 200 $env_cmd python scripts/gpu_timeline16t.py 16 $which > $OUT/tl_${which}_${tag}.log 2>&1
    echo "=== $which $cfg"; grep -v amdgpu.ids $OUT/tl_${which}_${tag}.log | grep -E "team|mid group 0|out group 0|segment sums|LN:|step total|wait at|gather|dst ids"
  done
done
timeout 400 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench_c3.log 2>&1; echo "rc=$?" >> $OUT/bench_c3.log; tail -n 2 $OUT/bench_c3.log | cut -c1-300
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_run.log 2>&1)
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c3_kernel_stats.csv \; 2>/dev/null
head -n 11 $OUT/c3_kernel_stats.csv | cut -c1-160
