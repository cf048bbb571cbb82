#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O/bench_all; bash tools/bench_all.sh > $O/bench_all/summary.log 2>&1; cat $O/bench_all/summary.log
FUZZ_ROUND2=1 FUZZ_ROUND3=1 timeout 900 python tests/fuzz_parity.py 200 3004 > $O/r03_fuzz_200.log 2>&1; tail -1 $O/r03_fuzz_200.log; grep FAIL $O/r03_fuzz_200.log | cut -c1-300
python -m pytest tests -m gpu -q -k "guard or compact" 2>&1 | tail -2
