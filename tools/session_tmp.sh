#!/bin/bash
# round 5 final measurement session (sources = HEAD)
cd "$(dirname "$0")/.."
export ROUND=r05
O=gpurun_out/r05_final; mkdir -p $O
bash tools/collect_all_profiles.sh > $O/collect.log 2>&1
bash tools/bench_all.sh > $O/bench_all.log 2>&1; cat $O/bench_all.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python tools/probes/split_shares.py config2 8 1,2 > $O/split_shares_config2.log 2>&1; tail -4 $O/split_shares_config2.log
timeout 300 python tools/probes/split_shares.py config4 8 1,2 > $O/split_shares_config4.log 2>&1; tail -3 $O/split_shares_config4.log
timeout 300 python tools/probes/shard_scaling.py config2 > $O/shard_scaling_config2.log 2>&1; tail -6 $O/shard_scaling_config2.log
( for smp in split guarded; do timeout 200 python tools/probes/multi_context_stress.py 320 200 3 200 $smp; done; timeout 200 python tools/probes/dense_shard_repro.py 2>&1 | tail -3 ) > $O/multi_context_stress.log 2>&1; cat $O/multi_context_stress.log
env -u RANK ADANERF_BENCH_DIST_BACKEND=gloo ADANERF_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2_one_device_gloo.json 2> $O/bench_gpus2.err; python - <<'PY'
import json
r=json.loads(open("gpurun_out/r05_final/bench_gpus2_one_device_gloo.json").read().strip().splitlines()[-1])
print("gpus 2 (one device, gloo):", r["value"], r["config"]["exchange"])
PY
du -sh gpurun_out/prof_r05_* $O
