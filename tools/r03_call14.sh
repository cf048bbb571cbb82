#!/bin/bash
# staged run-time-shaped kernels: tiles per sync / blocks per wave sweeps at widths 128 and 256, width 64 line
cd "$(dirname "$0")/.."
O=gpurun_out
L=tools/ablate_libs
mkdir -p $L/w128 $L/w256; mv $L/g128_*.so $L/w128/; mv $L/g256_*.so $L/w256/
run() { # dir workload
  for f in $L/$1/*.so; do
    ADANERF_LIB=$PWD/$f timeout 120 python bench.py --workload $2 --steps 20 --warmup 3 --no-cpu-baseline --no-speed-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$(basename $f .so)', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, 'shade frac', round(r['roofline']['frac'],3), 'sampling frac executed', round(r['sampling_roofline']['frac_executed'],3))"
  done
}
run w128 generic_6x128 > $O/r03_variants_generic_sweeps.log 2>&1
run w256 generic_5x256 >> $O/r03_variants_generic_sweeps.log 2>&1
for wl in generic_4x64 generic_6x128 generic_5x256; do
  python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-speed-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', 'shipped', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()}, 'shade frac', round(r['roofline']['frac'],3), 'sampling frac executed', round(r['sampling_roofline']['frac_executed'],3))"
done >> $O/r03_variants_generic_sweeps.log 2>&1
cat $O/r03_variants_generic_sweeps.log
ADANERF_LIB=$PWD/$L/w128/g128_tps2.so timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "generic or encoding or topolog" 2>&1 | tail -2
