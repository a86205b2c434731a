#!/bin/bash
OUT=gpurun_out/r02c; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 300 python scripts/gpu_timeline16.py 16 decoder > $OUT/timeline_dec.log 2>&1; tail -n 18 $OUT/timeline_dec.log
timeout 300 python scripts/gpu_timeline16.py 16 processor > $OUT/timeline_proc.log 2>&1; tail -n 18 $OUT/timeline_proc.log
timeout 600 python -m pytest tests/test_gpu_edge16.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "edge16 or bf16 or tiles" > $OUT/pytest_bf16.log 2>&1; tail -n 3 $OUT/pytest_bf16.log
timeout 400 python bench.py --config c3 --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $OUT/bench_c3.log 2>&1; tail -n 1 $OUT/bench_c3.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['other_kernels_ms'])"
