set -x
scripts/gpu_run.sh r06w5 "test:chain or layernorm or abi or version or loads"
cp gpurun_out/r06w5/pytest_gpu.log gpurun_out/r06w5/pytest_chain.log
scripts/gpu_run.sh r06w5 "test:backward or grad or train" bench:--mode+train+--precision+fp32+--steps+10+--warmup+3 bench:--mode+train+--precision+bf16x3+--steps+10+--warmup+3 proftrain:fp32
