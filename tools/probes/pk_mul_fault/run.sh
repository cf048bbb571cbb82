#!/bin/bash
# on the GPU box, from the repo root: every pkf_* library through the three-context isolation probe, REPS times each
cd "$(dirname "$0")/../../.."
for f in tools/ablate_libs/pkf_*.so; do
  for i in $(seq 1 ${REPS:-2}); do
    echo "== $(basename $f .so)"
    ADANERF_LIB=$PWD/$f timeout 300 python tools/probes/dense_stage_isolation.py ${M:-60} 2>&1 | grep -E "^render|^composite, |dense frames"
  done
done
