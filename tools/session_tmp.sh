#!/bin/bash
# session 25: two blocks per wave at width 128 shipped; generic tests, bench lines, fuzz over the run-time-shaped kinds
cd /root/repo
O=gpurun_out/r04_s25; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "generic or topolog or enc or norm or depth_cells or width" > $O/pytest_generic.log 2>&1; tail -2 $O/pytest_generic.log
for w in generic_6x128 generic_5x256 generic_4x64; do python bench.py --workload $w --steps 20 --no-cpu-baseline --no-speed-mode --no-exact-mode 2>/dev/null | tail -1 > $O/bench_$w.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_s25/bench_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), {k: round(v,3) for k,v in d['stage_ms_per_frame'].items()}, round(d['roofline']['frac'],3), round(d['roofline'].get('frac_of_sustained') or 0,3))
PY
FUZZ_KINDS=topo,enc,rsi,mult FUZZ_ROUND2=1 FUZZ_ROUND3=1 timeout 500 python tests/fuzz_parity.py 60 8101 > $O/fuzz_generic_60_seed8101.log 2>&1; tail -3 $O/fuzz_generic_60_seed8101.log
