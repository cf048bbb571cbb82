#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_s20; mkdir -p $O
for i in 1 2; do STEPS=20 BENCH_ARGS="--no-speed-mode --no-split-mode --no-sustained-probe --no-exact-mode --no-guarded-mode" bash tools/run_variants.sh; done | tee $O/variants.log
