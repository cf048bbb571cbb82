#!/bin/bash
# on the GPU box: bench every variant in tools/ablate_libs and print the per-stage ms; CHECK=1 also runs a quick parity
# subset of the GPU tests against each variant (ADANERF_LIB selects the library the Python host loads)
cd "$(dirname "$0")/.."
for f in tools/ablate_libs/*.so; do
  ADANERF_LIB=$PWD/$f timeout 120 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline ${BENCH_ARGS:---no-speed-mode --no-guarded-mode} 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $f .so)', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, round(r['config']['mean_samples_per_ray'],3), round(r['quality'].get('psnr_vs_oracle_db',0),1) if r['quality'] else '')"
  if [ "${CHECK:-0}" = "1" ]; then
    ADANERF_LIB=$PWD/$f timeout 300 python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "${CHECK_K:-frame_low_precision_psnr or shade_mlp_matches_oracle or render_is_deterministic}" 2>&1 | grep -E "^E  |passed|failed" | head -${CHECK_LINES:-6}
  fi
done
