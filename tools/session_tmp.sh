#!/bin/bash
# session 30: audit-fill with a guaranteed quarter quota as the hosts' default
cd /root/repo
O=gpurun_out/r04_s30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "guard or bench or smoke or frame" > $O/pytest_parity_subset.log 2>&1; tail -3 $O/pytest_parity_subset.log
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "guarded or config4 or config5 or config2" > $O/pytest_configs_subset.log 2>&1; tail -3 $O/pytest_configs_subset.log
python bench.py --no-cpu-baseline --no-speed-mode 2>/dev/null | tail -1 > $O/bench_fill.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_s30/bench_fill.json')); print(round(d['value'],1), d['stage_ms_per_frame'], d['config']['rays_refined_per_frame'], d['config']['guard']['rays_audited'], d['exact_mode']['value'], d['split_frame_mode']['value'])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
