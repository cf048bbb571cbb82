#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_final7; mkdir -p $O
FUZZ_ROUND2=1 FUZZ_ROUND3=1 timeout 1500 python tests/fuzz_parity.py 300 40404 > $O/fuzz_300_seed40404.log 2>&1; tail -1 $O/fuzz_300_seed40404.log; grep FAIL $O/fuzz_300_seed40404.log | cut -c1-260 | head -5; grep -c fragile $O/fuzz_300_seed40404.log
