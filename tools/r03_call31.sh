#!/bin/bash
# GPU session 31 of round 3 (final sources with the staged run-time-shaped kernels): full suite, kernel stats + PMC for the four
# single-GPU workloads, default bench line, generic / dense bench lines, smoke
cd "$(dirname "$0")/.."
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
export ADANERF_MEASURED_LOG=$PWD/$O/r03_measured31.log; rm -f $ADANERF_MEASURED_LOG
python -m pytest tests -m gpu -q > $O/r03_pytest_all31.log 2>&1; tail -5 $O/r03_pytest_all18.log
unset ADANERF_MEASURED_LOG
bash tools/collect_all_profiles.sh > $O/r03_collect_all.log 2>&1; tail -4 $O/r03_collect_all.log
for wl in config2 config3_dense config5_ndc generic_6x128; do cp $O/prof_r03_$wl/pmc_summary.json profiles/r03_pmc_summary_$wl.json; cp $O/prof_r03_$wl/kernel_stats.csv profiles/r03_rocprofv3_kernel_stats_$wl.csv; done
python bench.py > $O/r03_bench_final31.json 2> $O/r03_bench_final31.err; cut -c1-200 $O/r03_bench_final31.json
mkdir -p $O/r03_bench_all
python bench.py --workload config3_dense --steps 5 --warmup 2 --no-cpu-baseline --no-speed-mode > $O/r03_bench_all/config3_dense.json 2>/dev/null
for wl in generic_4x64 generic_6x128 generic_5x256; do python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-speed-mode > $O/r03_bench_all/${wl}_bf16.json 2>/dev/null; done
python bench.py --workload generic_6x128 --sampling fp32 --steps 20 --warmup 3 --no-cpu-baseline --no-speed-mode > $O/r03_bench_all/generic_6x128_bf16_fp32sampling.json 2>/dev/null
python - <<'PY'
import json, glob
for f in ["gpurun_out/r03_bench_final31.json"] + sorted(glob.glob("gpurun_out/r03_bench_all/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(r["value"], 2), {k: round(v, 3) for k, v in r["stage_ms_per_frame"].items()}, "frac", round(r["roofline"]["frac"], 3), "traffic", r["roofline"]["traffic"])
    except Exception as e:
        print(f, "unreadable", e)
PY
FUZZ_ROUND2=1 FUZZ_ROUND3=1 timeout 900 python tests/fuzz_parity.py 150 5301 > $O/r03_fuzz_final31_150.log 2>&1; tail -1 $O/r03_fuzz_final31_150.log; grep FAIL $O/r03_fuzz_final31_150.log | cut -c1-300
