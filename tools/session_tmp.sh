#!/bin/bash
# session 32: full GPU suite, profiles and bench lines on the final sources
cd /root/repo
O=gpurun_out/r04_s32; mkdir -p $O
ADANERF_MEASURED_LOG=$PWD/$O/parity_measured.log timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
tools/collect_all_profiles.sh > $O/collect.log 2>&1
tools/bench_all.sh > $O/bench_all.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/probes/split_shares.py config2 1,8 1,2 > $O/split_shares_config2_all_ranks.log 2>&1
cat $O/bench_all.log | head -8; grep -v amdgpu $O/split_shares_config2_all_ranks.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_s32/bench_default.json')); print(round(d['value'],1), d['stage_ms_per_frame'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('frac_of_sustained'), d['cpu_baseline']['value'], d['split_frame_mode']['value'], d['exact_mode']['value'], d['speed_mode']['value'])
PY
