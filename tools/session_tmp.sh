#!/bin/bash
# session 20: final profiles + bench lines on the frozen sources
cd /root/repo
O=gpurun_out/r04_s20; mkdir -p $O
tools/collect_all_profiles.sh > $O/collect.log 2>&1
tools/bench_all.sh > $O/bench_all.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tools/probes/mfma_peak > $O/mfma_peak.log 2>&1
tail -3 $O/collect.log; cat $O/bench_all.log; tail -c 3000 $O/bench_default.json
