#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
FUZZ_ONLY=36 FUZZ_ROUND2=1 FUZZ_ROUND3=1 FUZZ_KINDS=topo,enc,rsi,mult timeout 300 python tests/fuzz_parity.py 37 4102 > $O/r03_fuzz_case36_staged.log 2>&1; tail -12 $O/r03_fuzz_case36_staged.log | cut -c1-600
ADANERF_LIB=$PWD/tools/ablate_libs/g_direct.so FUZZ_ONLY=36 FUZZ_ROUND2=1 FUZZ_ROUND3=1 FUZZ_KINDS=topo,enc,rsi,mult timeout 300 python tests/fuzz_parity.py 37 4102 > $O/r03_fuzz_case36_direct.log 2>&1; tail -3 $O/r03_fuzz_case36_direct.log | cut -c1-600
