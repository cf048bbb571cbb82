O=$PWD/gpurun_out/r04_s6; mkdir -p $O; R=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "guard or compact or audit" > $O/pytest_guard.log 2>&1; tail -n 3 $O/pytest_guard.log
timeout 300 python bench.py --no-cpu-baseline --no-speed-mode > $O/bench_default.json 2> $O/bench_default.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/stats_mon -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-speed-mode --no-exact-mode --no-sustained-probe > $O/bench_mon.json 2> $O/bench_mon.err
cd $R; python - <<PY
import csv, collections, json
rows = list(csv.DictReader(open("$O/stats_mon/bench_kernel_trace.csv")))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0][:40]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items():
    big = [x for x in v if x > 0.25*max(v)]
    if "adanerf" in k: print("   %-42s calls %4d  big %3d  mean(big) %8.1f us  min %8.1f" % (k, len(v), len(big), sum(big)/len(big), min(big)))
r=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(round(r["value"],1), r["stage_ms_per_frame"], r["config"]["guard"], r["config"]["rays_refined_per_frame"], r["exact_mode"])
PY
