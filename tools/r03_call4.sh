#!/bin/bash
# GPU session 4 of round 3: interleave pinned with sched_group_barrier (two-block shading kernel, split sampling kernel)
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out
for v in base sb2 sb2sgb sb2sgbbp sb2sgbnr8 sb2sgbbw x3sgb x3sgb3 x3sgb8 both; do
  f=tools/ablate_libs/$v.so
  ADANERF_LIB=$R/$f timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-speed-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()})"
done > $O/r03_sgb.log 2>&1
cat $O/r03_sgb.log
for v in sb2sgb x3sgb; do
ADANERF_LIB=$R/tools/ablate_libs/$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "shade_mlp_matches_oracle or frame_low_precision_psnr or render_is_deterministic or full_size or sample_mlp_matches or fused_selection" 2>&1 | tail -2
done
