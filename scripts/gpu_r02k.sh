#!/bin/bash
OUT=gpurun_out/r02k; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
bash scripts/gpu_pmc_c3.sh r02k/pmc_c3 > $OUT/pmc_c3.log 2>&1; python - <<'PY'
import json
d=json.load(open("gpurun_out/r02k/pmc_c3/pmc_c3_edge16.json"))
for k,v in d.items():
    print(k, {a: (round(b/1e6,1) if b>1e4 else round(b,3)) for a,b in v.items() if a.startswith("hbm") or a in ("mfma_busy_frac","GRBM_GUI_ACTIVE")})
PY
for b in 2 4 8 16; do
  timeout 300 python bench.py --config c3 --batch $b --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>&1 | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('batch', $b, round(d['value'],1), 'f/s', round(d['ms_per_step'],3), 'ms', d['roofline'].get('other_kernels_ms'), round(d['roofline']['launch_ms'],3))"
done
