#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
for lib in "" "$PWD/tools/ablate_libs/g_round3_before_staging.so"; do
  echo "lib: ${lib:-shipped}"
  ADANERF_LIB=$lib FUZZ_ONLY=88 FUZZ_ROUND2=1 FUZZ_ROUND3=1 timeout 300 python tests/fuzz_parity.py 89 5301 2>&1 | grep -E "^worst ray|^case" | cut -c1-400
done > $O/r03_fuzz_case88_ab.log 2>&1; cat $O/r03_fuzz_case88_ab.log
