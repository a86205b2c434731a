#!/bin/bash
OUT=gpurun_out/r02x; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
cat > /tmp/wide_fwd.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
print(bench.run_wide(torch.device("cuda:0"), steps=4, warmup=2)["ms_per_step"])
PY
rm -rf /tmp/prof_w && mkdir -p /tmp/prof_w
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -o gw -- python /tmp/wide_fwd.py > $GRAFT_REPO_ROOT/$OUT/rocprof_wide.log 2>&1)
find /tmp/prof_w -name "*kernel_stats*" -exec cp {} $OUT/wide_kernel_stats.csv \; 2>/dev/null
head -n 14 $OUT/wide_kernel_stats.csv | cut -c1-200
tail -n 2 $OUT/rocprof_wide.log | cut -c1-200
