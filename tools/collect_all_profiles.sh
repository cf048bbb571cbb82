#!/bin/bash
# on the GPU box: kernel stats + PMC passes (tools/collect_profiles.sh) for the single-GPU workloads whose
# roofline.traffic bench.py reports -> gpurun_out/prof_${ROUND:-r05}_<workload>/
cd "$(dirname "$0")/.."
STEPS=20 PROF_DIR=prof_${ROUND:-r05}_config2 WORKLOAD=config2 BENCH_ARGS="" tools/collect_profiles.sh > gpurun_out/collect_${ROUND:-r05}_config2.log 2>&1
STEPS=4 PROF_DIR=prof_${ROUND:-r05}_config3_dense WORKLOAD=config3_dense BENCH_ARGS="--workload config3_dense" tools/collect_profiles.sh > gpurun_out/collect_${ROUND:-r05}_config3.log 2>&1
STEPS=20 PROF_DIR=prof_${ROUND:-r05}_config5_ndc WORKLOAD=config5_ndc BENCH_ARGS="--workload config5_ndc --precision fp16" tools/collect_profiles.sh > gpurun_out/collect_${ROUND:-r05}_config5.log 2>&1
STEPS=20 PROF_DIR=prof_${ROUND:-r05}_generic_6x128 WORKLOAD=generic_6x128 BENCH_ARGS="--workload generic_6x128" tools/collect_profiles.sh > gpurun_out/collect_${ROUND:-r05}_generic.log 2>&1
for d in gpurun_out/prof_${ROUND:-r05}_*; do rm -rf $d/stats/*.db $d/pmc_*/*.db $d/stats/bench_kernel_trace.csv $d/pmc_*/pmc_kernel_trace.csv; done
du -sh gpurun_out/prof_${ROUND:-r05}_*
