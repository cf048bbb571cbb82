#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_s16; mkdir -p $O
for wl in generic_4x64 generic_6x128; do echo "== $wl"; for i in 1 2; do STEPS=20 BENCH_ARGS="--workload $wl --no-speed-mode --no-split-mode --no-sustained-probe --no-exact-mode --no-guarded-mode" bash tools/run_variants.sh; done; done | tee $O/variants.log
