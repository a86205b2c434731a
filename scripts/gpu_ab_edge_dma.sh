#!/bin/bash
# A/B on one box (tuning build): DMA piece schedule of edge_kernel - GW_EDGE_DMA6=0 (two per K-step over the first four K-steps)
# against GW_EDGE_DMA6=1 (2, 1, 1, 2, 1, 1 over six).  usage: scripts/gpu_ab_edge_dma.sh TAG
TAG=${1:-ab_dma}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GW_TUNING=1 GW_AUTO_GRAPH=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
for REP in 1 2; do for V in 0 1; do
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && GW_EDGE_DMA6=$V timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $R/bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $R/$OUT/rocprof_$V.log 2>&1)
  find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/k_$V.csv \; 2>/dev/null
  echo "== GW_EDGE_DMA6=$V (rep $REP): $(grep -o '"value": [0-9.]*' $OUT/rocprof_$V.log | tail -n 1)"
  python - $OUT/k_$V.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "edge_kernel<" in n:
        print("   %-60s calls %4s avg %8.1f us  min %8.1f us" % (n[n.find("::") + 2:][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done; done
