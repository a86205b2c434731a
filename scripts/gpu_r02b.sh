#!/bin/bash
# Round 2, call b: new bf16 edge kernels (8-wave resident kernel, layer-1 kernel, edge tiles) + hunt for the memory fault.
OUT=gpurun_out/r02b; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_edge16.py tests/test_gpu_parity.py -m gpu -q -s -x --timeout 600 -p no:cacheprovider -k "edge16 or bf16 or tiles or integration" > $OUT/pytest_bf16.log 2>&1
echo "pytest bf16 rc=$?" >> $OUT/pytest_bf16.log; tail -n 25 $OUT/pytest_bf16.log
timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -q -s --timeout 600 -p no:cacheprovider -k "c3" > $OUT/pytest_c3.log 2>&1
echo "pytest c3 rc=$?" >> $OUT/pytest_c3.log; tail -n 12 $OUT/pytest_c3.log
timeout 400 python bench.py --config c3 --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $OUT/bench_c3.log 2>&1; echo "bench c3 rc=$?" >> $OUT/bench_c3.log; tail -n 2 $OUT/bench_c3.log | cut -c1-1500
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c3.log 2>&1)
find /tmp/prof -name "*kernel_stats*" -exec cp {} $OUT/c3_kernel_stats.csv \; 2>/dev/null
head -n 14 $OUT/c3_kernel_stats.csv | cut -c1-200
# the memory access fault of call a: every tensor its own allocation, every launch blocking -> the faulting launch has a Python stack
PYTORCH_NO_CUDA_MEMORY_CACHING=1 HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 900 python -X faulthandler -m pytest tests/test_gpu_round2.py -m gpu -q -s -x --timeout 800 -p no:cacheprovider -k "graphcast or use_checkpointing or two_step or frozen or flat" > $OUT/pytest_fault.log 2>&1
echo "pytest fault-hunt rc=$?" >> $OUT/pytest_fault.log; grep -v "site-packages\|dist-packages\|^$" $OUT/pytest_fault.log | head -n 60 | cut -c1-300
