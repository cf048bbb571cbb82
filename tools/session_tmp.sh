#!/bin/bash
# session 29: full GPU suite, profiles and bench lines on the current sources
cd /root/repo
O=gpurun_out/r04_s29; mkdir -p $O
ADANERF_MEASURED_LOG=$PWD/$O/parity_measured.log timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
tools/collect_all_profiles.sh > $O/collect.log 2>&1
tools/bench_all.sh > $O/bench_all.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
adanerf_amd/bin/mfma_peak > $O/mfma_peak.log 2>&1
cat $O/bench_all.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_s29/bench_default.json')); print(round(d['value'],1), d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('frac_of_sustained'), d['cpu_baseline']['value'], d['split_frame_mode']['value'], d['exact_mode']['value'])
PY
