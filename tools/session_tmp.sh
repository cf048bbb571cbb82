#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_s11; mkdir -p $O
ADANERF_LIB_A=$PWD/tools/ablate_libs/base.so timeout 600 python tools/probes/compare_libs.py > $O/compare_libs_clamp.log 2>&1; grep -c "identical True, raw shading outputs identical True" $O/compare_libs_clamp.log; grep -c "False" $O/compare_libs_clamp.log
for i in 1 2; do for v in base shipped; do
  L=$PWD/tools/ablate_libs/base.so; [ $v = shipped ] && L=$PWD/adanerf_amd/lib/libadanerf_hip.so
  for wl in config2 generic_5x256 generic_6x128 generic_4x64; do ADANERF_LIB=$L timeout 200 python bench.py --workload $wl --steps 20 --no-cpu-baseline --no-speed-mode --no-split-mode --no-guarded-mode --no-sustained-probe 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $wl', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, round(r['roofline']['frac'],4))"; done; done; done | tee $O/clamp_ab.log
