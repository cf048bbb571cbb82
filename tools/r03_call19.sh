#!/bin/bash
# selection fused into the run-time-shaped split-precision sampling kernel: parity, fuzz (generic kinds), the three widths
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "generic or encoding or topolog or coarse or keep_oracle or stage" > $O/r03_gen_fused_tests.log 2>&1; tail -3 $O/r03_gen_fused_tests.log
FUZZ_ROUND2=1 FUZZ_ROUND3=1 FUZZ_KINDS=topo,enc,rsi timeout 600 python tests/fuzz_parity.py 80 4103 > $O/r03_fuzz_gen_fused.log 2>&1; tail -1 $O/r03_fuzz_gen_fused.log; grep FAIL $O/r03_fuzz_gen_fused.log | cut -c1-300 | head -5
for wl in generic_4x64 generic_6x128 generic_5x256; do
  python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-speed-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, 'shade frac', round(r['roofline']['frac'],3), 'compact GB/s', r['hbm_stages']['compact_GBps'])"
done > $O/r03_generic_fused_selection.log 2>&1
cat $O/r03_generic_fused_selection.log
