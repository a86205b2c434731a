#!/bin/bash
# A/B of the row-split node update on ONE box (tuning build): kernel statistics of the 1 degree, batch 2 forward with the knobs
#   GW_NODE_RS=0|1 (64-column kernels | row-split form), GW_RS_TUNE bit 0 (no L2 prefetch of the weight stream), + phase clocks.
# usage: scripts/gpu_ab_rs.sh TAG
TAG=${1:-ab_rs}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GW_TUNING=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
for CFG in c2 c2x3; do
  for V in "GW_NODE_RS=0" "GW_NODE_RS=1 GW_RS_TUNE=1" "GW_NODE_RS=1 GW_RS_TUNE=0"; do
    N=$(echo "$V" | tr -c 'a-zA-Z0-9' '_')
    rm -rf /tmp/prof && mkdir -p /tmp/prof
    (cd /tmp && env $V timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $R/$OUT/rocprof_${CFG}_$N.log 2>&1)
    find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/${CFG}_${N}_kernel_stats.csv \; 2>/dev/null
    echo "== $CFG $V: $(grep -o '"value": [0-9.]*' $OUT/rocprof_${CFG}_$N.log | tail -n 1) forecasts/s under rocprofv3"
    python - $OUT/${CFG}_${N}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "node_rs" in n or "chain_kernel<64, true, 2, 16, 16, 0, false, true>" in n or "chainx3_kernel<8, true, 2, 16, 16, 0, false, true" in n:
        print("   %-70s calls %4s avg %8.1f us  min %8.1f us  %5s %%" % (n[n.find("::") + 2:][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Percentage"]))
PY
  done
done
for P in fp32 bf16x3; do for T in 0 1; do GW_RS_TUNE=$T python scripts/gpu_timeline_rs.py 2 $P 2>&1 | grep -v "amdgpu.ids"; done; done > $OUT/rs_timeline.log
cat $OUT/rs_timeline.log
