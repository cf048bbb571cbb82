#!/bin/bash
cd "$(dirname "$0")/.."
export ROUND=r05
rm -rf gpurun_out/prof_r05_config2 gpurun_out/prof_r05_config3_dense gpurun_out/prof_r05_config5_ndc gpurun_out/prof_r05_generic_6x128
O=gpurun_out/r05_final9; mkdir -p $O
bash tools/collect_all_profiles.sh > $O/collect.log 2>&1
for wl in config2 config3_dense config5_ndc generic_6x128; do cp gpurun_out/prof_r05_$wl/pmc_summary.json profiles/r05_pmc_summary_$wl.json; done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
r=json.loads(open("gpurun_out/r05_final9/bench_default.json").read().strip().splitlines()[-1])
print(r["value"], r["stage_ms_per_frame"], r["roofline"]["frac"], r["roofline"]["traffic"], r["roofline"]["traffic_source"], r["guarded_mode"]["value"], r["guarded_mode"]["ahead_of_the_headline"], r["speed_mode"]["value"], r["cpu_baseline"]["value"])
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "shade_mlp or frame" 2>&1 | tail -2
