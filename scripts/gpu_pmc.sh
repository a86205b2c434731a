#!/bin/bash
# PMC passes for the bench workload (separate runs per counter group, kernel-trace only - no sys/hip tracing).
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/counters_list.txt 2>&1
run_pass () {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/run_$name.log 2>&1
  echo "rc=$?" >> $GRAFT_REPO_ROOT/$OUT/run_$name.log
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    # keep only our chain kernels, aggregated per (kernel, counter)
    python - "$f" > $GRAFT_REPO_ROOT/$OUT/pmc_$name.txt <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        k = r.get("Kernel_Name", "")
        m = re.search(r"(chain_kernel|edge_kernel)<[^>]*>", k)
        if not m: continue
        key = (m.group(0), r.get("Grid_Size"), r["Counter_Name"])
        agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
for (k, g, c), (v, n) in sorted(agg.items()):
    print(f"{k}\tgrid={g}\t{c}\tmean={v/n:.6g}\tn={n}")
PY
  fi
}
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAVES
run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
[ "${PMC_LDS:-1}" = "1" ] && run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
ls -la $GRAFT_REPO_ROOT/$OUT
tail -n 60 $GRAFT_REPO_ROOT/$OUT/pmc_sq.txt
