#!/bin/bash
OUT=gpurun_out/r02h; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/rocprof_train.log 2>&1)
find /tmp/prof -name "*kernel_stats*" -exec cp {} $OUT/train_kernel_stats.csv \; 2>/dev/null
head -n 32 $OUT/train_kernel_stats.csv | cut -c1-150
python - $OUT/train_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("launches per step:", sum(int(r["Calls"]) for r in rows) / 5.0, " GPU ms per step:", sum(float(r["TotalDurationNs"]) for r in rows) / 5e6)
PY
tail -n 1 $OUT/rocprof_train.log | cut -c1-300
