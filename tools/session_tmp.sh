#!/bin/bash
# session 21: LDS bias table in the run-time-shaped shading kernel; vectorised mask walk of refine_list_kernel
cd /root/repo
O=gpurun_out/r04_s21; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -m gpu > $O/pytest_configs.log 2>&1; tail -3 $O/pytest_configs.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "guard or refine or generic or topolog" > $O/pytest_parity_subset.log 2>&1; tail -3 $O/pytest_parity_subset.log
for w in generic_6x128 generic_5x256 generic_4x64; do python bench.py --workload $w --steps 20 --no-cpu-baseline --no-speed-mode --no-exact-mode 2>/dev/null | tail -1 > $O/bench_$w.json; done
python bench.py --steps 30 --no-cpu-baseline --no-speed-mode --no-exact-mode 2>/dev/null | tail -1 > $O/bench_config2.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_s21/bench_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), {k: round(v,3) for k,v in d['stage_ms_per_frame'].items()}, round(d['roofline']['frac'],3), round(d['config']['mean_samples_per_ray'],3))
PY
