#!/bin/bash
# GPU session 2 of round 3: guarded-selection tests with the calibrated band, shading-kernel variants (two sample blocks per
# wave), the hand-scheduled ablations re-run from the fixed experiment guard, band probe, bench.
cd "$(dirname "$0")/.."
O=gpurun_out
export ADANERF_MEASURED_LOG=$PWD/$O/r03_measured2.log; rm -f $ADANERF_MEASURED_LOG
python -m pytest tests -m gpu -q -x -k "guard or compact_guarded" > $O/r03_pytest_guard2.log 2>&1; tail -3 $O/r03_pytest_guard2.log
for v in base sb2 sb2nr8 sb2rs4; do
  f=tools/ablate_libs/$v.so
  ADANERF_LIB=$PWD/$f timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-speed-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()})"
  ADANERF_LIB=$PWD/$f timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "shade_mlp_matches_oracle or frame_low_precision_psnr or render_is_deterministic" 2>&1 | grep -E "^E  |passed|failed" | head -4
done > $O/r03_variants_sb2.log 2>&1
cat $O/r03_variants_sb2.log
for v in base hs hs_a1 hs_a2 hs_a3 hs_a4 hs_a8 hs_a14 hs_a15 hs_a16 base_a15; do
  f=tools/ablate_libs/$v.so
  ADANERF_LIB=$PWD/$f timeout 120 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-speed-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()})"
done > $O/r03_ablate_handsched.log 2>&1
cat $O/r03_ablate_handsched.log
python tools/probes/guard_band.py config2 config5_ndc > $O/r03_guard_band2.log 2>&1; grep eps_requested $O/r03_guard_band2.log | cut -c1-400
python bench.py --sampling guarded > $O/r03_bench_guarded2.json 2> $O/r03_bench_guarded2.err; cut -c1-600 $O/r03_bench_guarded2.json
python -m pytest tests -m gpu -q > $O/r03_pytest_all2.log 2>&1; tail -4 $O/r03_pytest_all2.log
