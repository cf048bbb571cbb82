#!/bin/bash
# staged run-time-shaped kernels with the position encoding parked in LDS: parity, fuzz, the three widths
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "generic or encoding or topolog or coarse" > $O/r03_gen_staged_tests.log 2>&1; tail -3 $O/r03_gen_staged_tests.log
FUZZ_ROUND2=1 FUZZ_ROUND3=1 FUZZ_KINDS=topo,enc,rsi,mult timeout 600 python tests/fuzz_parity.py 80 4102 > $O/r03_fuzz_gen_staged.log 2>&1; tail -1 $O/r03_fuzz_gen_staged.log; grep FAIL $O/r03_fuzz_gen_staged.log | cut -c1-300 | head -5
for wl in generic_4x64 generic_6x128 generic_5x256; do
  python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-speed-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', 'shipped', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, 'shade frac', round(r['roofline']['frac'],3), 'sampling frac executed', round(r['sampling_roofline']['frac_executed'],3))"
done > $O/r03_generic_staged_stash.log 2>&1
cat $O/r03_generic_staged_stash.log
