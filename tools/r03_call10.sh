#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
python -m pytest tests -m gpu -q -k "bench or dense or fp32_matches" 2>&1 | tail -3
python tools/probes/frames_in_flight.py config2 > $O/r03_frames_in_flight.log 2>&1; cat $O/r03_frames_in_flight.log | tail -4
python bench.py --workload config3_dense --steps 5 --warmup 2 --no-cpu-baseline --no-speed-mode > $O/r03_bench_dense_final.json 2> $O/r03_bench_dense_final.err; python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03_bench_dense_final.json").read().strip().splitlines()[-1])
print("dense", round(r["value"], 2), {k: round(v, 3) for k, v in r["stage_ms_per_frame"].items()}, "frac", round(r["roofline"]["frac"], 3), "traffic", r["roofline"]["traffic"], r["hbm_stages"]["compact_GBps"])
PY
