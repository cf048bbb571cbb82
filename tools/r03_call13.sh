#!/bin/bash
# run-time-shaped 16-bit kernels with LDS-staged weight tiles: parity tests, fuzz on the new features, A/B against the direct form
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "generic or encoding or topolog or coarse" > $O/r03_gen_staged_tests.log 2>&1; tail -3 $O/r03_gen_staged_tests.log
FUZZ_ROUND2=1 FUZZ_ROUND3=1 FUZZ_KINDS=topo,enc,rsi,mult timeout 600 python tests/fuzz_parity.py 60 4101 > $O/r03_fuzz_gen_staged.log 2>&1; tail -1 $O/r03_fuzz_gen_staged.log; grep FAIL $O/r03_fuzz_gen_staged.log | cut -c1-300 | head -5
STEPS=20 BENCH_ARGS="--workload generic_6x128 --no-speed-mode" tools/run_variants.sh > $O/r03_variants_generic_staged.log 2>&1; cat $O/r03_variants_generic_staged.log
