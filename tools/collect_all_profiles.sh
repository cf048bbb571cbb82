#!/bin/bash
# on the GPU box: kernel stats + PMC passes (tools/collect_profiles.sh) for the single-GPU workloads whose
# roofline.traffic bench.py reports -> gpurun_out/prof_r03_<workload>/
cd "$(dirname "$0")/.."
STEPS=20 PROF_DIR=prof_r03_config2 WORKLOAD=config2 BENCH_ARGS="" tools/collect_profiles.sh > gpurun_out/collect_r03_config2.log 2>&1
STEPS=4 PROF_DIR=prof_r03_config3_dense WORKLOAD=config3_dense BENCH_ARGS="--workload config3_dense" tools/collect_profiles.sh > gpurun_out/collect_r03_config3.log 2>&1
STEPS=20 PROF_DIR=prof_r03_config5_ndc WORKLOAD=config5_ndc BENCH_ARGS="--workload config5_ndc --precision fp16" tools/collect_profiles.sh > gpurun_out/collect_r03_config5.log 2>&1
STEPS=20 PROF_DIR=prof_r03_generic_6x128 WORKLOAD=generic_6x128 BENCH_ARGS="--workload generic_6x128" tools/collect_profiles.sh > gpurun_out/collect_r03_generic.log 2>&1
for d in gpurun_out/prof_r03_*; do rm -rf $d/stats/*.db $d/pmc_*/*.db $d/stats/bench_kernel_trace.csv $d/pmc_*/pmc_kernel_trace.csv; done
du -sh gpurun_out/prof_r03_*
