set -x
scripts/gpu_run.sh r06w2 "test:tests/test_gpu_train_kernels.py"
cp gpurun_out/r06w2/pytest_gpu.log gpurun_out/r06w2/pytest_train_kernels.log
scripts/gpu_run.sh r06w2 "test:backward or grad or train or split or chain or abi or version" bench:--mode+train+--precision+bf16x3+--steps+10+--warmup+3 bench:--mode+train+--precision+fp32+--steps+10+--warmup+3 proftrain:bf16x3
