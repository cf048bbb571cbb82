#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_final10; mkdir -p $O
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
r=json.loads(open("gpurun_out/r05_final10/bench_default.json").read().strip().splitlines()[-1])
print(r["value"], r["stage_ms_per_frame"], r["roofline"]["frac"], r["roofline"]["traffic"], r["guarded_mode"]["value"], r["guarded_mode"]["ahead_of_the_headline"], r["speed_mode"]["value"], r["cpu_baseline"], r["quality"])
PY
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "config2 or config4 or rows" 2>&1 | tail -2
