#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_s5; mkdir -p $O
echo "== --sampling split" > $O/variants.log
STEPS=20 BENCH_ARGS="--sampling split --no-speed-mode --no-split-mode --no-sustained-probe --no-exact-mode" bash tools/run_variants.sh >> $O/variants.log 2>&1
echo "== guarded (default)" >> $O/variants.log
STEPS=20 BENCH_ARGS="--no-speed-mode --no-split-mode --no-sustained-probe --no-exact-mode" bash tools/run_variants.sh >> $O/variants.log 2>&1
cat $O/variants.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "ray_table" 2>&1 | tail -5
