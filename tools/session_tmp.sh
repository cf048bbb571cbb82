#!/bin/bash
# round 5 session 1: probes, SLP / split_pack A/B, bench self-launch tests
cd "$(dirname "$0")/.."
O=gpurun_out/r05_s1; mkdir -p $O
( timeout 60 /tmp/cvt 2>/dev/null; hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probes/cvt_clamp_probe.hip -o /tmp/cvt_clamp && timeout 60 /tmp/cvt_clamp ) > $O/cvt_clamp.log 2>&1
timeout 300 adanerf_amd/bin/mfma_peak > $O/mfma_peak.log 2>&1
echo "== config2 (guarded headline + exact mode)" > $O/variants.log
for f in tools/ablate_libs/*.so; do
  ADANERF_LIB=$PWD/$f timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-speed-mode --no-split-mode --no-sustained-probe 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $f .so)', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, 'exact', round(r['exact_mode']['value'],1), round(r['exact_mode']['sample_mlp_ms'],3))" >> $O/variants.log 2>&1
done
for wl in generic_6x128 generic_5x256 generic_4x64 config5_ndc; do
  echo "== $wl" >> $O/variants.log
  for v in base noslp sp2 sp2_noslp; do
    EXTRA=""; [ $wl = config5_ndc ] && EXTRA="--precision fp16"
    ADANERF_LIB=$PWD/tools/ablate_libs/$v.so timeout 200 python bench.py --workload $wl $EXTRA --steps 20 --warmup 5 --no-cpu-baseline --no-speed-mode --no-split-mode --no-sustained-probe --no-exact-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()})" >> $O/variants.log 2>&1
  done
done
cat $O/variants.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -m gpu -k "bench" 2>&1 | tail -15 > $O/pytest_bench.log
cat $O/pytest_bench.log; tail -12 $O/mfma_peak.log; cat $O/cvt_clamp.log
