#!/bin/bash
cd "$(dirname "$0")/.."
export ROUND=r05
STEPS=20 PROF_DIR=prof_r05_config2_guarded WORKLOAD=config2_guarded BENCH_ARGS="--sampling guarded" tools/collect_profiles.sh > gpurun_out/collect_r05_config2_guarded.log 2>&1
rm -rf gpurun_out/prof_r05_config2_guarded/stats/*.db gpurun_out/prof_r05_config2_guarded/pmc_*/*.db gpurun_out/prof_r05_config2_guarded/stats/bench_kernel_trace.csv gpurun_out/prof_r05_config2_guarded/pmc_*/pmc_kernel_trace.csv
head -8 gpurun_out/prof_r05_config2_guarded/kernel_stats.csv | cut -c1-160
