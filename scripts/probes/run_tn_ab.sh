export GW_TUNING=1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/tn_build.log 2>&1 || { tail gpurun_out/tn_build.log; exit 1; }
run() { echo "== $*"; env "$@" python scripts/probes/gemm_tn_x3_probe.py 2>&1 | grep -v amdgpu.ids; }
{
run GW_TN_TARGET=1024
run GW_TN_TARGET=512 GW_TN_CAP=16384
run GW_TN_TARGET=512 GW_TN_CAP=16384 GW_TN_MIN=512
run GW_TN_TARGET=768 GW_TN_CAP=16384
run GW_TN_TARGET=512 GW_TN_CAP=16384 GW_TN_X3_TUNE=1
run GW_TN_TARGET=256 GW_TN_CAP=16384
} | tee gpurun_out/tn_ab2.log
