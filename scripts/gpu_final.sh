#!/bin/bash
# End-of-round evidence in one call (profiles/rNN_* are copies of what this writes under gpurun_out/TAG):
#   scripts/gpu_final.sh TAG   ->  env, GPU tests, smoke, the driver's bench command, rocprofv3 kernel stats of c2 / c3, the
#   phase clocks of the team kernels (decoder / processor), the C3 PMC passes.
TAG=${1:-final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ date; rocm-smi --showproductname --showmeminfo vram 2>/dev/null | head -20; lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"; python -c "import torch; print(torch.__version__, torch.version.hip)"; } > $OUT/env.log 2>&1
scripts/gpu_run.sh $TAG build test smoke bench
scripts/gpu_run.sh $TAG prof:c2 prof:c3 > $OUT/prof_summary.log 2>&1
bash scripts/gpu_pmc_c3.sh $TAG > $OUT/pmc_c3_summary.log 2>&1
# phase clocks are compiled into tuning builds only: rebuild (the box is scratch; nothing after this uses the product build)
export GW_TUNING=1
python -c "import __graft_entry__ as g; g.build()" > $OUT/build_tuning.log 2>&1
python scripts/gpu_timeline16t.py 16 decoder > $OUT/team_timeline_decoder.log 2>&1
python scripts/gpu_timeline16t.py 16 processor > $OUT/team_timeline_processor.log 2>&1
tail -n 12 $OUT/team_timeline_decoder.log
