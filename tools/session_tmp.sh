#!/bin/bash
cd "$(dirname "$0")/.."
export ROUND=r05
rm -rf gpurun_out/prof_r05_* gpurun_out/r05_bench_all
O=gpurun_out/r05_final5; mkdir -p $O
ADANERF_LIB_A=$PWD/tools/ablate_libs/base.so timeout 600 python tools/probes/compare_libs.py > $O/compare_libs.log 2>&1; grep -c "identical True, raw shading outputs identical True" $O/compare_libs.log; grep -c False $O/compare_libs.log
ADANERF_MEASURED_LOG=$PWD/$O/parity_measured.log timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
bash tools/collect_all_profiles.sh > $O/collect.log 2>&1
for wl in config2 config3_dense config5_ndc generic_6x128; do cp gpurun_out/prof_r05_$wl/pmc_summary.json profiles/r05_pmc_summary_$wl.json; done
bash tools/bench_all.sh > $O/bench_all.log 2>&1; head -3 $O/bench_all.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
r=json.loads(open("gpurun_out/r05_final5/bench_default.json").read().strip().splitlines()[-1])
print(r["value"], r["stage_ms_per_frame"], r["roofline"]["frac"], r["roofline"]["traffic"], r["roofline"]["traffic_source"], r["guarded_mode"]["value"], r["guarded_mode"]["ahead_of_the_headline"], r["speed_mode"]["value"], r["cpu_baseline"]["value"])
PY
