#!/bin/bash
# A/B of the edge-update implementations: parity tests on the new path, bench on both, decoder timeline of the new one.
TAG=${1:-ab}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
export GW_TUNING=1  # the env knobs exist only in tuning builds (-DGW_TUNING)
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || echo BUILD FAILED
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -n 15 $OUT/pytest_gpu.log
for impl in 1 0; do
  GW_EDGE_IMPL=$impl timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_impl$impl.log 2>&1
  python - $OUT/bench_impl$impl.log "impl=$impl" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); r = d["roofline"]
        print(sys.argv[2], "fc/s=%.1f ms/step=%.2f dec_ms=%.3f exec_frac=%.3f proc_edge=%.3f enc_edge=%.3f" % (d["value"], d["ms_per_step"], r["launch_ms"], r["executed_frac"], r["other_kernels_ms"]["processor_edge"], r["other_kernels_ms"]["encoder_edge"]))
        break
else:
    print(sys.argv[2], "FAILED"); print(open(sys.argv[1]).read()[-1500:])
PY
done
timeout 300 python scripts/gpu_timeline.py $OUT/timeline_dec.npy 1 > $OUT/timeline.log 2>&1; tail -2 $OUT/timeline.log
