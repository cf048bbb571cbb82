#!/bin/bash
# Round 6, final session on ONE box: GPU test suite, default bench line, every workload, kernel stats + PMC passes, multi-context stress, N = 8 projection.
cd "$(dirname "$0")/.."; O=gpurun_out/r06_final; mkdir -p $O
python -m pytest tests -m gpu -q > $O/r06_pytest_gpu_final.log 2>&1; tail -3 $O/r06_pytest_gpu_final.log
python bench.py > $O/r06_bench_default.json 2> $O/r06_bench_default.err; python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06_final/r06_bench_default.json").read().strip().splitlines()[-1])
print("default:", round(r["value"], 1), "FPS", {k: round(v, 3) for k, v in r["stage_ms_per_frame"].items()}, "frac", round(r["roofline"]["frac"], 3), "of sustained", round(r["roofline"].get("frac_of_sustained") or 0, 3),
      "guarded", round(r["guarded_mode"]["value"], 1), r["guarded_mode"]["auto_choice"], "speed", round(r["speed_mode"]["value"], 1), "cpu", round(r["cpu_baseline"]["value"], 3), "box", r["roofline"]["box"])
print("quality:", r["quality"])
PY
ROUND=r06 bash tools/bench_all.sh > $O/r06_bench_all.log 2>&1; cat $O/r06_bench_all.log
ROUND=r06 bash tools/collect_all_profiles.sh > $O/collect.log 2>&1; tail -5 $O/collect.log
python tools/probes/multi_context_stress.py 800 800 8 40 split > $O/r06_multi_context_stress.log 2>&1; tail -2 $O/r06_multi_context_stress.log
python tools/probes/split_shares.py config2 1,8 1,2 > $O/r06_split_shares_config2.log 2>&1; tail -4 $O/r06_split_shares_config2.log
python tools/probes/split_shares.py config4 1,8 1,2 > $O/r06_split_shares_config4.log 2>&1; tail -4 $O/r06_split_shares_config4.log
