#!/bin/bash
# L2 request counters of one bench configuration (own --pmc passes, kernel-trace only):  scripts/gpu_pmc_l2.sh TAG CONFIG
#   -> gpurun_out/TAG/pmc_icache_CONFIG.json: per kernel instantiation and grid, average per launch of TCC_REQ / TCC_HIT / TCC_MISS (summed over
#      the channels) and TCP_TCC_READ_REQ; the decoder edge update's weight stream is L2 -> LDS traffic that HBM counters do not see.
TAG=${1:-pmc}; CFG=${2:-c2x3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
run_pass () {
  name=$1; shift
  rm -rf /tmp/pmcic_$name
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcic_$name -o p -- python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $R/$OUT/run_ic_${CFG}_$name.log 2>&1
  echo "rc=$?" >> $R/$OUT/run_ic_${CFG}_$name.log
  f=$(find /tmp/pmcic_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" /tmp/rawic_${CFG}_$name.csv
}
run_pass ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run_pass if SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
run_pass grbm GRBM_GUI_ACTIVE
python - "$R/$OUT" "$CFG" <<'PY'
import csv, sys, os, re, json, collections
out, cfg = sys.argv[1], sys.argv[2]
rx = re.compile(r"(chainx3_kernel|chain_kernel|edge_kernel)<[^>]*>")
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for name in ("ic", "if", "grbm"):
    p = f"/tmp/rawic_{cfg}_{name}.csv"
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        m = rx.search(r.get("Kernel_Name", ""))
        if not m: continue
        a = agg[f"{m.group(0)} grid={r.get('Grid_Size')}"][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
res = {}
for k, cs in agg.items():
    d = {c: v / n for c, (v, n) in cs.items()}
    d["launches_seen"] = max(n for _, n in cs.values())
    if "GRBM_GUI_ACTIVE" in d: d["kernel_cycles"] = d["GRBM_GUI_ACTIVE"] / 8.0
    res[k] = d
json.dump(res, open(os.path.join(out, f"pmc_icache_{cfg}.json"), "w"), indent=1, sort_keys=True)
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("SQC_ICACHE_REQ", 0)):
    print(k[:100], {c: int(v) for c, v in d.items()})
PY
