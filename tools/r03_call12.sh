#!/bin/bash
# generic split-precision sampling kernel: parity tests + the 6 x 128 workload with both sampling engines
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "generic or encoding or topolog" > $O/r03_gen_split_tests.log 2>&1; tail -3 $O/r03_gen_split_tests.log
for smp in guarded fp32; do
  python bench.py --workload generic_6x128 --sampling $smp --steps 30 --warmup 5 --no-speed-mode --cpu-budget 1 2>/dev/null | tail -1 > $O/r03_bench_generic_$smp.json
  python -c "
import json; r=json.load(open('$O/r03_bench_generic_$smp.json')); print('$smp', round(r['value'],1), {k:round(v,3) for k,v in r['stages_ms'].items()} if 'stages_ms' in r else '', r['sampling_roofline']['frac_executed'], r['roofline']['frac'])"
done
