#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_final4; mkdir -p $O
timeout 300 python tools/probes/pass_variance.py 2>&1 | tail -5 | tee $O/pass_variance.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
r=json.loads(open("gpurun_out/r05_final4/bench_default.json").read().strip().splitlines()[-1])
print(r["value"], r["stage_ms_per_frame"], r["roofline"]["frac"], r["roofline"]["traffic"], r["guarded_mode"]["value"], r["guarded_mode"]["ahead_of_the_headline"], r["speed_mode"]["value"], r["split_frame_mode"]["value"], r["cpu_baseline"]["value"])
PY
