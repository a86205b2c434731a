#!/bin/bash
# Sweep an env knob over bench.py (no CPU baseline) and print value / kernel times.
# usage: scripts/gpu_tune.sh TAG VAR v1 v2 ...
TAG=$1; VAR=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
export GW_TUNING=1  # the env knobs exist only in tuning builds (-DGW_TUNING)
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || echo BUILD FAILED
for v in "$@"; do
  env $VAR=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_${VAR}_$v.log 2>&1
  python - "$OUT/bench_${VAR}_$v.log" "$VAR=$v" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); r = d["roofline"]
        print(sys.argv[2], "fc/s=%.1f ms/step=%.2f dec_ms=%.3f exec_frac=%.3f proc_edge=%.3f enc_edge=%.3f" % (d["value"], d["ms_per_step"], r["launch_ms"], r["executed_frac"], r["other_kernels_ms"]["processor_edge"], r["other_kernels_ms"]["encoder_edge"]))
        break
else:
    print(sys.argv[2], "FAILED"); print(open(sys.argv[1]).read()[-800:])
PY
done
if [ -n "$TIMELINE" ]; then python scripts/gpu_timeline.py $OUT/timeline_dec.npy 1; fi
