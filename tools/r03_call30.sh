#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, r['config'].get('guard'), 'speed', r['speed_mode']['value'] if r.get('speed_mode') else None)"
