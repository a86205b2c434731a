#!/bin/bash
# Copy the judged summaries of a scripts/gpu_final.sh run (gpurun_out/TAG, scratch) into profiles/ (tracked): scripts/collect_profiles.sh TAG rNN
TAG=${1:-r06}; R=${2:-r06}; S=gpurun_out/$TAG; D=profiles
cp $S/env.log $D/${R}_env.log
cp $S/pytest_gpu.log $D/${R}_pytest_gpu.log
cp $S/smoke.log $D/${R}_smoke.log
cp $S/bench___steps_20___warmup_5_.log $D/${R}_bench_default.log
for c in c2 c2x3 c3; do cp $S/${c}_kernel_stats.csv $D/${R}_${c}_kernel_stats.csv; done
for p in fp32 bf16x3; do [ -f $S/train_${p}_kernel_stats.csv ] && cp $S/train_${p}_kernel_stats.csv $D/${R}_train_${p}_kernel_stats.csv; done
for c in c2 c2x3; do
  [ -f $S/pmc_${c}_dominant.json ] && cp $S/pmc_${c}_dominant.json $D/${R}_pmc_${c}.json
  [ -f $S/pmc_${c}.json ] && cp $S/pmc_${c}.json $D/${R}_pmc_${c}_all_kernels.json
done
for f in x3_timeline_b2 x3_timeline_b16 rs_timeline_b2; do [ -f $S/$f.log ] && cp $S/$f.log $D/${R}_$f.log; done
ls -la $D/${R}_* | wc -l
