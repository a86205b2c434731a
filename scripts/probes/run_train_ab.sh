set -x
scripts/gpu_run.sh r06w4 "test:chain or layernorm or abi or version or loads"
cp gpurun_out/r06w4/pytest_gpu.log gpurun_out/r06w4/pytest_chain.log
scripts/gpu_run.sh r06w4 "test:tests/test_gpu_split.py" bench:--mode+train+--precision+bf16x3+--steps+10+--warmup+3 proftrain:bf16x3
