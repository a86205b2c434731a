#!/bin/bash
# round 3, call q: chain16 8 waves x 1 group against 4 x 2 (tuning build: GW_CHAIN16_NW)
OUT=gpurun_out/r03q; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for nw in 4 8; do
  GW_CHAIN16_NW=$nw timeout 120 python scripts/probes/chain16_probe.py 2>&1 | grep rows | sed "s/$/ NW=$nw/" | tee -a $OUT/probe.log
  GW_CHAIN16_NW=$nw timeout 300 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench_c3_nw$nw.log 2>&1
  tail -n 1 $OUT/bench_c3_nw$nw.log | cut -c1-400
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && GW_CHAIN16_NW=$nw timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gw -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/rocprof_c3_nw$nw.log 2>&1)
  find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/c3_kernel_stats_nw$nw.csv \; 2>/dev/null
  head -n 9 $OUT/c3_kernel_stats_nw$nw.csv | cut -c1-170
done
GW_CHAIN16_NW=8 timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "bf16 or round3 or c3 or narrow or guards" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -n 6 $OUT/pytest.log
