#!/bin/bash
# PMC passes of one bench configuration (separate runs per counter group, kernel-trace only - no sys/hip tracing):
#   scripts/gpu_pmc_cfg.sh TAG CONFIG [KERNEL_REGEX]      ->  gpurun_out/TAG/pmc_CONFIG.json  (per kernel instantiation and grid)
#                                                           + gpurun_out/TAG/pmc_CONFIG_dominant.json: the edge-update instantiation with the
#                                                             largest grid (the decoder edge update) in the flat form bench.pmc_traffic() reads
#                                                             (copied to profiles/rNN_pmc_CONFIG.json by scripts/gpu_final.sh)
# PMC_CMD (optional): the bench.py arguments of the profiled run instead of "--config CONFIG --steps 3 --warmup 1 --no-cpu-baseline
# --no-extra" - e.g. the training step: PMC_CMD="--mode train --precision bf16x3 --steps 2 --warmup 1" scripts/gpu_pmc_cfg.sh TAG train_x3
# "(bwd_chainx3_kernel|gemm_tn_x3_kernel)<[^>]*>"
# HBM bytes: FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) and WRITE_SIZE, both in KiB; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /
# 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); LDS array utilisation = SQ_LDS_IDX_ACTIVE / 256 CUs / (GRBM_GUI_ACTIVE / 8).
TAG=${1:-pmc}; CFG=${2:-c2}; RX=${3:-"(chainx3_kernel|chain_kernel|edge_kernel|chain16_kernel|node_rs3?_kernel|edge16[a-z_0-9]*kernel)<[^>]*>"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=${PMC_CMD:-"--config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-extra"}
cd /tmp
run_pass () {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  GW_AUTO_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $R/bench.py $CMD > $R/$OUT/run_${CFG}_$name.log 2>&1
  echo "rc=$?" >> $R/$OUT/run_${CFG}_$name.log
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" /tmp/raw_${CFG}_$name.csv
}
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
[ "${PMC_LDS:-1}" = "1" ] && run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
python - "$R/$OUT" "$CFG" "$RX" <<'PY'
import csv, sys, os, re, json, collections
out, cfg, rx = sys.argv[1], sys.argv[2], re.compile(sys.argv[3])
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for name in ("sq", "grbm", "fetch", "write", "lds"):
    p = f"/tmp/raw_{cfg}_{name}.csv"
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        m = rx.search(r.get("Kernel_Name", ""))
        if not m: continue
        key = f"{m.group(0)} grid={r.get('Grid_Size')}"
        a = agg[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
res = {}
for k, cs in agg.items():
    d = {c: v / n for c, (v, n) in cs.items()}
    d["launches_seen"] = max(n for _, n in cs.values())
    if "FETCH_SIZE" in d: d["hbm_read_bytes"] = 2.0 * d["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in d: d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024.0
    g = d.get("GRBM_GUI_ACTIVE")
    if g:
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d: d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (g / 8.0)
        if "SQ_LDS_IDX_ACTIVE" in d: d["lds_array_frac"] = d["SQ_LDS_IDX_ACTIVE"] / 256.0 / (g / 8.0)
        d["kernel_cycles"] = g / 8.0
    res[k] = d
json.dump(res, open(os.path.join(out, f"pmc_{cfg}.json"), "w"), indent=1, sort_keys=True)
# the dominant kernel of the forward = the edge-update kernel instantiation launched with the largest grid (decoder edge update)
edge = [(float(k.split("grid=")[1]), k) for k in res if re.match(r"(edge_kernel<|chainx3_kernel<8, true, 3, 16, 16, 1,|edge16t_kernel<)", k)]
if edge:
    _, k = max(edge)
    d = dict(res[k]); d["kernel"] = k.split(" grid=")[0]; d["grid_threads"] = int(float(k.split("grid=")[1])); d["config"] = cfg
    json.dump(d, open(os.path.join(out, f"pmc_{cfg}_dominant.json"), "w"), indent=1, sort_keys=True)
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1].get("launches_seen", 1)):
    print(k[:110], {c: (round(v, 3) if v < 100 else int(v)) for c, v in d.items() if c in ("mfma_busy_frac", "lds_array_frac", "kernel_cycles", "hbm_read_bytes", "hbm_write_bytes", "launches_seen", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM")})
PY
