set -x
scripts/gpu_run.sh r06w3 "test:tests/test_gpu_train_kernels.py"
cp gpurun_out/r06w3/pytest_gpu.log gpurun_out/r06w3/pytest_train_kernels.log
scripts/gpu_run.sh r06w3 "test:tests/test_gpu_split.py" bench:--mode+train+--precision+bf16x3+--steps+10+--warmup+3 proftrain:bf16x3
