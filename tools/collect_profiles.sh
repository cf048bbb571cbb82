#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root.  Collects, for the bench.py default workload:
#   gpurun_out/prof/kernel_stats.csv        rocprofv3 --kernel-trace --stats
#   gpurun_out/prof/pmc_<set>/...           one rocprofv3 --pmc pass per counter set (never combined with
#                                           other trace domains, as the pool requires)
#   gpurun_out/prof/pmc_summary.json        per-kernel means (tools/summarize_pmc.py)
# Copy what should be judged into profiles/ afterwards.
set -u
R=$PWD
OUT=$R/gpurun_out/${PROF_DIR:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-speed-mode --no-exact-mode --no-guarded-mode --no-split-mode --no-sustained-probe ${BENCH_ARGS:-}"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py $ARGS > $OUT/stats.log 2>&1
cp $OUT/stats/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  N=$(echo $P | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_$N -o pmc -- python $R/bench.py $ARGS > $OUT/pmc_$N.log 2>&1
done
cd $R
PMC_WORKLOAD=${WORKLOAD:-config2} BENCH_ARGS="${BENCH_ARGS:-}" python tools/summarize_pmc.py $OUT > $OUT/pmc_summary.json
cat $OUT/kernel_stats.csv | head -8
head -c 1500 $OUT/pmc_summary.json
