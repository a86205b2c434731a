#!/bin/bash
# End-of-round evidence in one call (profiles/rNN_* are copies of what this writes under gpurun_out/TAG):
#   scripts/gpu_final.sh TAG   ->  env, GPU tests, smoke, the driver's bench command, rocprofv3 kernel stats of c2 / c2x3 / c3, the
#   PMC passes of c2 (fp32 headline) and c2x3 (split mode), the phase clocks of the split kernels.
# STEPS (env, optional) selects a subset: "test smoke bench prof pmc clocks" (default: all).
TAG=${1:-final}
STEPS=${STEPS:-"test smoke bench prof pmc clocks"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ date; rocm-smi --showproductname --showmeminfo vram 2>/dev/null | head -20; lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"; python -c "import torch; print(torch.__version__, torch.version.hip)"; } > $OUT/env.log 2>&1
scripts/gpu_run.sh $TAG build
case " $STEPS " in *" test "*) scripts/gpu_run.sh $TAG test;; esac
case " $STEPS " in *" smoke "*) scripts/gpu_run.sh $TAG smoke; scripts/gpu_run.sh $TAG py:scripts/probes/graph_rccl_probe.py | tail -n 6;; esac
case " $STEPS " in *" bench "*) scripts/gpu_run.sh $TAG bench | cut -c1-6000; scripts/gpu_run.sh $TAG py:scripts/probes/cold_parts_probe.py | tail -n 3;; esac
case " $STEPS " in *" prof "*) scripts/gpu_run.sh $TAG prof:c2 prof:c2x3 prof:c3 proftrain:fp32 proftrain:bf16x3 > $OUT/prof_summary.log 2>&1; tail -n 40 $OUT/prof_summary.log | cut -c1-200;; esac
case " $STEPS " in *" pmc "*)
  PMC_LDS=0 bash scripts/gpu_pmc_cfg.sh $TAG c2 > $OUT/pmc_c2_summary.log 2>&1
  bash scripts/gpu_pmc_cfg.sh $TAG c2x3 > $OUT/pmc_c2x3_summary.log 2>&1
  head -n 3 $OUT/pmc_c2_summary.log | cut -c1-600; head -n 3 $OUT/pmc_c2x3_summary.log | cut -c1-600;; esac
case " $STEPS " in *" clocks "*)
  # phase clocks are compiled into tuning builds only: rebuild (the box is scratch; nothing after this uses the product build)
  export GW_TUNING=1
  python -c "import __graft_entry__ as g; g.build()" > $OUT/build_tuning.log 2>&1
  for W in decoder processor node dechead nodeenc; do python scripts/gpu_timeline_x3.py 2 $W 2>&1 | grep -v "amdgpu.ids"; done > $OUT/x3_timeline_b2.log
  for W in decoder processor; do python scripts/gpu_timeline_x3.py 16 $W 2>&1 | grep -v "amdgpu.ids"; done > $OUT/x3_timeline_b16.log
  for P in fp32 bf16x3; do python scripts/gpu_timeline_rs.py 2 $P 2>&1 | grep -v "amdgpu.ids"; done > $OUT/rs_timeline_b2.log
  tail -n 10 $OUT/x3_timeline_b2.log;; esac
