#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
for rep in 1 2; do for f in tools/ablate_libs/*.so; do
  ADANERF_LIB=$PWD/$f timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-speed-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $f .so)', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, 'refined', r['sampling_roofline']['rays_refined_per_frame'], 'guard', r['config'].get('guard'))"
done; done > $O/r03_variants_pair_xchg.log 2>&1; cat $O/r03_variants_pair_xchg.log
