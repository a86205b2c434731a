#!/bin/bash
# A/B of the split weight-gradient GEMM (gemm_tn_x3_kernel, csrc/gw_train.hip) on a tuning build, at the row counts of the 1 degree
# training step (scripts/probes/gemm_tn_x3_probe.py).  From the repo root on the GPU box: bash scripts/gpu_ab_gemm_tn_x3.sh
# -> gpurun_out/tn_ab.log (profiles/r06_gemm_tn_x3_ab.log is a run of this script).
#   GW_TN_X3_TUNE bits: 1 = no atomics, 2 = launch-order slabs (no XCD grouping), 4 = no MFMAs
#   GW_TN_TARGET / GW_TN_CAP / GW_TN_MIN: workgroups aimed at, row cap and minimum of a slab (product: 512 / 16384 / 256)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GW_TUNING=1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/tn_build.log 2>&1 || { tail gpurun_out/tn_build.log; exit 1; }
run() { echo "== $*"; env "$@" python scripts/probes/gemm_tn_x3_probe.py 2>&1 | grep -v amdgpu.ids; }
{
run GW_TN_X3_TUNE=0
run GW_TN_X3_TUNE=2
run GW_TN_X3_TUNE=1
run GW_TN_X3_TUNE=4
run GW_TN_X3_TUNE=5
run GW_TN_TARGET=1024 GW_TN_CAP=4096
run GW_TN_TARGET=768
run GW_TN_TARGET=256
run GW_TN_MIN=512
} | tee gpurun_out/tn_ab.log
