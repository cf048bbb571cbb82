#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
FUZZ_ROUND2=1 FUZZ_ROUND3=1 timeout 900 python tests/fuzz_parity.py 150 5301 > $O/r03_fuzz_final26_150.log 2>&1; tail -1 $O/r03_fuzz_final26_150.log; grep FAIL $O/r03_fuzz_final26_150.log | cut -c1-300
FUZZ_ROUND2=1 FUZZ_ROUND3=1 FUZZ_KINDS=pdf,pdf_ce,coarse_fine,cf_ndc timeout 900 python tests/fuzz_parity.py 80 5302 > $O/r03_fuzz_classic_80.log 2>&1; tail -1 $O/r03_fuzz_classic_80.log; grep FAIL $O/r03_fuzz_classic_80.log | cut -c1-300
