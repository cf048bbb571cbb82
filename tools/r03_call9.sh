#!/bin/bash
# GPU session 9 of round 3 (final sources): full suite, kernel stats + PMC for the three single-GPU workloads, the default bench line
# with roofline.traffic from those summaries, dense bench line
cd "$(dirname "$0")/.."
O=gpurun_out
export ADANERF_MEASURED_LOG=$PWD/$O/r03_measured9.log; rm -f $ADANERF_MEASURED_LOG
python -m pytest tests -m gpu -q > $O/r03_pytest_all9.log 2>&1; tail -5 $O/r03_pytest_all9.log
unset ADANERF_MEASURED_LOG
bash tools/collect_all_profiles.sh > $O/r03_collect_all.log 2>&1; tail -3 $O/r03_collect_all.log
for wl in config2 config3_dense config5_ndc; do cp $O/prof_r03_$wl/pmc_summary.json profiles/r03_pmc_summary_$wl.json; cp $O/prof_r03_$wl/kernel_stats.csv profiles/r03_rocprofv3_kernel_stats_$wl.csv; done
grep -h "^{" $O/prof_r03_config2/stats.log | tail -1 > $O/r03_bench_config2_under_rocprof.json
python bench.py > $O/r03_bench_final.json 2> $O/r03_bench_final.err; cut -c1-200 $O/r03_bench_final.json
python bench.py --workload config3_dense --steps 5 --warmup 2 --no-cpu-baseline --no-speed-mode > $O/r03_bench_dense_final.json 2>/dev/null; python - <<'PY'
import json
for f in ("r03_bench_final", "r03_bench_dense_final"):
    r = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(r["value"], 2), {k: round(v, 3) for k, v in r["stage_ms_per_frame"].items()}, "frac", round(r["roofline"]["frac"], 3), "traffic", r["roofline"]["traffic"], r["roofline"]["traffic_source"])
PY
