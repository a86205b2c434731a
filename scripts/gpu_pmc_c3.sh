#!/bin/bash
# PMC passes for the bf16 configuration (c3: 1 degree, batch 16): HBM traffic and MFMA activity of the edge16 kernels, per
# position in the forward (launch order of edge16_kernel within a step: first processor block, blocks 1..8, decoder).
TAG=${1:-pmc_c3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
run_pass () {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/run_$name.log 2>&1
  echo "rc=$?" >> $GRAFT_REPO_ROOT/$OUT/run_$name.log
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$OUT/raw_$name.csv
}
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python - $GRAFT_REPO_ROOT/$OUT <<'PY'
import csv, sys, os, re, json, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for name in ("fetch", "write", "sq", "grbm"):
    p = os.path.join(out, f"raw_{name}.csv")
    if not os.path.exists(p): continue
    rows = list(csv.DictReader(open(p)))
    # order of dispatches; position of each edge16_kernel launch within its forward (10 per forward)
    by_disp = collections.OrderedDict()
    for r in rows:
        by_disp.setdefault(int(r["Dispatch_Id"]), []).append(r)
    pos = collections.Counter()
    for d in sorted(by_disp):
        rs = by_disp[d]
        k = rs[0]["Kernel_Name"]
        m = re.search(r"(edge16t_kernel<[^>]*>|edge16p_kernel<[^>]*>|edge16_kernel<[^>]*>|edge16_l1_kernel|chain16_kernel<[^>]*>)", k)
        if not m: continue
        kind = m.group(1)
        i = pos[kind]; pos[kind] += 1
        # per forward (round 3): the team kernel's gather form runs twice (encoder edge update, then decoder), its DMA form and
        # the layer-1 kernel once per processor block 1..8, the lock-step kernel once (first processor block); the chain16
        # kernels (node-side MLPs) are averaged per instantiation
        # (round 4: the decoder runs edge16t_kernel<true, true, false, true> - segment-aligned tiles - the encoder the form
        #  without them, processor blocks 1..8 the layer-1 kernel + edge16p_kernel<e' out>)
        # (both run the gather form on segment-aligned tiles since the encoder's split runs: launch 0, 2, 4, ... of that
        #  instantiation is the encoder's edge update, the odd ones are the decoder's)
        if kind.startswith("edge16t_kernel<true"):
            label = "encoder" if i % 2 == 0 else "decoder"
        elif kind.startswith(("edge16t_kernel", "edge16p_kernel")) or kind == "edge16_l1_kernel":
            label = "blocks1-8"
        elif kind.startswith("edge16_kernel"):
            label = "block0"
        else:
            label = "node side"
        for r in rs:
            key = (kind, label, r["Counter_Name"])
            a = res[key]
            a["sum"] = a.get("sum", 0.0) + float(r["Counter_Value"]); a["n"] = a.get("n", 0) + 1
summary = {}
for (kind, label, c), a in sorted(res.items()):
    summary.setdefault(f"{kind} [{label}]", {})[c] = a["sum"] / a["n"]
for k, v in summary.items():
    if "FETCH_SIZE" in v: v["hbm_read_bytes (FETCH_SIZE KiB x 1024 x 2, gfx950 correction)"] = v["FETCH_SIZE"] * 2048.0
    if "WRITE_SIZE" in v: v["hbm_write_bytes (WRITE_SIZE KiB x 1024)"] = v["WRITE_SIZE"] * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0)
json.dump(summary, open(os.path.join(out, "pmc_c3.json"), "w"), indent=1)
print(json.dumps({k: {c: v[c] for c in v if c.startswith(('hbm', 'mfma'))} for k, v in summary.items()}, indent=1))
PY
rm -f $GRAFT_REPO_ROOT/$OUT/raw_*.csv
