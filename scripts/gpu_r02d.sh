#!/bin/bash
OUT=gpurun_out/r02d; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_edge16.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "post_products or forecaster or bf16 or zero_parameters or integration or edge16 or tiles or graphcast_wrapper" > $OUT/pytest.log 2>&1; tail -n 6 $OUT/pytest.log
for c in c2 c3; do
timeout 400 python bench.py --config $c --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $OUT/bench_$c.log 2>&1; tail -n 1 $OUT/bench_$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['other_kernels_ms'])"
done
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c3.log 2>&1)
find /tmp/prof -name "*kernel_stats*" -exec cp {} $OUT/c3_kernel_stats.csv \; 2>/dev/null
head -n 14 $OUT/c3_kernel_stats.csv | cut -c1-180
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c2.log 2>&1)
find /tmp/prof -name "*kernel_stats*" -exec cp {} $OUT/c2_kernel_stats.csv \; 2>/dev/null
head -n 10 $OUT/c2_kernel_stats.csv | cut -c1-180
