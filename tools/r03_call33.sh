#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
for lib in "" "$PWD/tools/ablate_libs/g_direct.so"; do
  echo "lib: ${lib:-shipped}"
  ADANERF_LIB=$lib FUZZ_ONLY=67 FUZZ_ROUND2=1 FUZZ_ROUND3=1 timeout 300 python tests/fuzz_parity.py 68 6001 2>&1 | grep -E "^worst ray|^case|raw\| max" | cut -c1-420
done > $O/r03_fuzz_case67_ab.log 2>&1; cat $O/r03_fuzz_case67_ab.log
