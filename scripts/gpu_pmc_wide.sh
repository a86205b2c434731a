#!/bin/bash
# MFMA activity of the wide path's GEMM (gemm_nt_kernel) at the decoder's edge-level product (453 600 x 1024 x 1024).
OUT=gpurun_out/pmc_wide; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
cat > /tmp/gemm_one.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from graph_weather_amd import wide
x = torch.randn(453600, 1024, device="cuda:0"); w = torch.randn(1024, 1024, device="cuda:0") / 32
for _ in range(3):
    y = wide.linear_forward(x, w, None, True)
torch.cuda.synchronize()
PY
cd /tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$name -o p -- python /tmp/gemm_one.py > $GRAFT_REPO_ROOT/$OUT/run_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep "gemm_nt_kernel" "$f" | head -40 > $GRAFT_REPO_ROOT/$OUT/raw_$name.csv
  [ -n "$f" ] && head -1 "$f" > $GRAFT_REPO_ROOT/$OUT/header_$name.csv
done
python - $GRAFT_REPO_ROOT/$OUT <<'PY'
import csv, sys, os, json, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
    hdr = open(os.path.join(out, f"header_{name}.csv")).readline().strip().replace('"', '').split(",")
    for row in csv.reader(open(os.path.join(out, f"raw_{name}.csv"))):
        r = dict(zip(hdr, row))
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in agg.items()}
if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "GRBM_GUI_ACTIVE" in res:
    res["mfma_busy_frac"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (res["GRBM_GUI_ACTIVE"] / 8.0)
res["kernel"] = "gemm_nt_kernel<true>, 453600 x 1024 -> 1024"
json.dump(res, open(os.path.join(out, "pmc_wide_gemm.json"), "w"), indent=1)
print(json.dumps(res))
PY
