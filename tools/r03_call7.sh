#!/bin/bash
# GPU session 7 of round 3: final profiles (kernel stats + PMC for the three single-GPU workloads), one bench line per
# configuration, fuzz harness with the round-2 and round-3 features, shard-scaling projection
cd "$(dirname "$0")/.."
O=gpurun_out
bash tools/collect_all_profiles.sh > $O/r03_collect_all.log 2>&1; tail -3 $O/r03_collect_all.log
mkdir -p $O/bench_all; bash tools/bench_all.sh > $O/bench_all/summary.log 2>&1; cat $O/bench_all/summary.log
FUZZ_ROUND2=1 FUZZ_ROUND3=1 timeout 600 python tests/fuzz_parity.py 150 3003 > $O/r03_fuzz_150.log 2>&1; tail -2 $O/r03_fuzz_150.log; grep -c FAIL $O/r03_fuzz_150.log
python tools/probes/shard_scaling.py config2 0.2 1,2,4,8 > $O/r03_shard_scaling_config2_full.log 2>&1; tail -3 $O/r03_shard_scaling_config2_full.log
python tools/probes/shard_scaling.py config5_ndc 0.2 1,8 fp16 > $O/r03_shard_scaling_config5.log 2>&1; tail -1 $O/r03_shard_scaling_config5.log
