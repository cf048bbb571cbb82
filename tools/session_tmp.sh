#!/bin/bash
# round 5 final measurement session (sources = HEAD)
cd "$(dirname "$0")/.."
export ROUND=r05
rm -rf gpurun_out/prof_r05_* gpurun_out/r05_bench_all
O=gpurun_out/r05_final2; mkdir -p $O
ADANERF_MEASURED_LOG=$PWD/$O/parity_measured.log timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4 > $O/smoke.log; cat $O/smoke.log
bash tools/collect_all_profiles.sh > $O/collect.log 2>&1
bash tools/bench_all.sh > $O/bench_all.log 2>&1; cat $O/bench_all.log
timeout 300 python tools/probes/split_shares.py config2 8 1,2 > $O/split_shares_config2.log 2>&1; tail -2 $O/split_shares_config2.log
timeout 300 python tools/probes/split_shares.py config4 8 1,2 > $O/split_shares_config4.log 2>&1; tail -2 $O/split_shares_config4.log
( for smp in split guarded; do timeout 200 python tools/probes/multi_context_stress.py 320 200 3 200 $smp; done; timeout 200 python tools/probes/dense_shard_repro.py 2>&1 | tail -3 ) > $O/multi_context_stress.log 2>&1; cat $O/multi_context_stress.log
timeout 900 python tests/fuzz_parity.py 150 30303 > $O/fuzz_150_seed30303.log 2>&1; tail -1 $O/fuzz_150_seed30303.log; grep FAIL $O/fuzz_150_seed30303.log | head -3
du -sh gpurun_out/prof_r05_* $O
