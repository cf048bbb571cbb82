#!/bin/bash
# session 27: bench.py with sub-shares per GPU (default for N > 1) and split_frame_mode at N = 1
cd /root/repo
O=gpurun_out/r04_s27; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "test_bench" > $O/pytest_bench.log 2>&1; tail -5 $O/pytest_bench.log
python bench.py --no-cpu-baseline --no-speed-mode 2>$O/bench.err | tail -1 > $O/bench_default.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_s27/bench_default.json')); print(round(d['value'],1), d['stage_ms_per_frame'], d['exact_mode']['value'], d['split_frame_mode'])
PY
python tools/probes/split_shares.py config4 1,8 1,2 > $O/split_shares_config4.log 2>&1; grep -v amdgpu.ids $O/split_shares_config4.log
python tools/probes/split_shares.py config5_ndc 1,8 1,2 fp16 > $O/split_shares_config5.log 2>&1; grep -v amdgpu.ids $O/split_shares_config5.log
