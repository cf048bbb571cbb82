#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
export ADANERF_MEASURED_LOG=$PWD/$O/r03_measured23.log; rm -f $ADANERF_MEASURED_LOG
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "guard" 2>&1 | tail -15
grep guard_self $ADANERF_MEASURED_LOG
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-speed-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, r['config'].get('guard'))"
