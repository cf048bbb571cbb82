#!/bin/bash
# GPU session 6 of round 3: full suite (16-bit generic engine, wave-granular partial round), shard-scaling projection, generic bench line
cd "$(dirname "$0")/.."
O=gpurun_out
export ADANERF_MEASURED_LOG=$PWD/$O/r03_measured6.log; rm -f $ADANERF_MEASURED_LOG
python -m pytest tests -m gpu -q > $O/r03_pytest_all6.log 2>&1; tail -12 $O/r03_pytest_all6.log
python tools/probes/shard_scaling.py config2 0.2 1,8 > $O/r03_shard_scaling_config2.log 2>&1; tail -2 $O/r03_shard_scaling_config2.log
python tools/probes/shard_scaling.py config4 0.1 1,8 > $O/r03_shard_scaling_config4.log 2>&1; tail -1 $O/r03_shard_scaling_config4.log
python bench.py --workload generic_6x128 --steps 10 > $O/r03_bench_generic_bf16.json 2> $O/r03_bench_generic.err; cut -c1-250 $O/r03_bench_generic_bf16.json
python bench.py --workload generic_6x128 --steps 10 --precision fp32 --no-cpu-baseline > $O/r03_bench_generic_fp32.json 2>> $O/r03_bench_generic.err; cut -c1-250 $O/r03_bench_generic_fp32.json
