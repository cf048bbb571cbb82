#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
FUZZ_ROUND2=1 FUZZ_ROUND3=1 timeout 900 python tests/fuzz_parity.py 150 3005 > $O/r03_fuzz_final_150.log 2>&1; tail -1 $O/r03_fuzz_final_150.log; grep FAIL $O/r03_fuzz_final_150.log | cut -c1-300
for pt in 64,4 32,8 128,2 16,16 64,2 32,4; do
  ADANERF_CPU_PT=$pt python bench.py --steps 5 --warmup 2 --no-speed-mode --cpu-budget 6 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['cpu_baseline']; print('$pt', round(c['value'],4), c['cores'], c['sample'][:120])"
done > $O/r03_cpu_pt.log 2>&1; cat $O/r03_cpu_pt.log
