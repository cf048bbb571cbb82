#!/bin/bash
# on the GPU box: bench every variant in tools/ablate_libs and print the per-stage ms
cd "$(dirname "$0")/.."
for f in tools/ablate_libs/*.so; do
  ADANERF_LIB=$PWD/$f timeout 120 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $f .so)', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, round(r['config']['mean_samples_per_ray'],3), round(r['quality'].get('psnr_vs_oracle_db',0),1) if r['quality'] else '')"
done
