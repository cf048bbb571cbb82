#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
for lib in "" "$PWD/tools/ablate_libs/g_round3_before_staging.so"; do
  ADANERF_LIB=$lib FUZZ_ONLY=19 FUZZ_ROUND2=1 FUZZ_ROUND3=1 FUZZ_KINDS=topo,enc,rsi timeout 300 python tests/fuzz_parity.py 20 4103 2>&1 | grep -E "^worst ray|^case|raw\| max" | cut -c1-400
done > $O/r03_fuzz_case19_ab.log 2>&1; cat $O/r03_fuzz_case19_ab.log
