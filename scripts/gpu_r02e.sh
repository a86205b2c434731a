#!/bin/bash
OUT=gpurun_out/r02e; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "streams or forecaster_matches or full_size" > $OUT/pytest.log 2>&1; tail -n 6 $OUT/pytest.log
for st in 1 2; do
timeout 400 python bench.py --config c2 --streams $st --steps 30 --warmup 5 --no-extra --no-cpu-baseline > $OUT/bench_c2_s$st.log 2>&1; tail -n 1 $OUT/bench_c2_s$st.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 streams $st', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['other_kernels_ms'])"
done
for st in 1 2 4; do
timeout 400 python bench.py --config c2 --batch 8 --streams $st --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $OUT/bench_b8_s$st.log 2>&1; tail -n 1 $OUT/bench_b8_s$st.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 B=8 streams $st', d['value'], d['ms_per_step'])"
done
for st in 1 2; do
timeout 400 python bench.py --config c3 --streams $st --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $OUT/bench_c3_s$st.log 2>&1; tail -n 1 $OUT/bench_c3_s$st.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 streams $st', d['value'], d['ms_per_step'])"
done
