#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_s4; mkdir -p $O
REPS=2 M=60 bash tools/probes/pk_mul_fault/run.sh > $O/pk_mul_fault.log 2>&1
cat $O/pk_mul_fault.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "ray_table or ray_features" 2>&1 | tail -5 > $O/pytest_ray.log; cat $O/pytest_ray.log
timeout 900 python tests/fuzz_parity.py 80 10101 > $O/fuzz_80_seed10101.log 2>&1; tail -4 $O/fuzz_80_seed10101.log; grep -c "fragile" $O/fuzz_80_seed10101.log; grep FAIL $O/fuzz_80_seed10101.log | head
