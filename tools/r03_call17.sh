#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
for f in tools/ablate_libs/*.so; do
  ADANERF_LIB=$PWD/$f timeout 120 python bench.py --workload generic_6x128 --steps 20 --warmup 3 --no-cpu-baseline --no-speed-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $f .so)', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, 'shade frac', round(r['roofline']['frac'],3))"
done > $O/r03_variants_generic_occ.log 2>&1; cat $O/r03_variants_generic_occ.log
