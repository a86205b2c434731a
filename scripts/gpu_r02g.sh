#!/bin/bash
OUT=gpurun_out/r02g; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_edge16.py tests/test_gpu_narrow.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "deterministic or without_norm or integration or edge_cases or edge16 or narrow" > $OUT/pytest.log 2>&1; tail -n 8 $OUT/pytest.log | cut -c1-300
timeout 400 python bench.py --mode train --steps 5 --warmup 2 > $OUT/bench_train.log 2>&1; tail -n 1 $OUT/bench_train.log | cut -c1-700
