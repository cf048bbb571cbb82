#!/bin/bash
# Builds experiment variants of the library into tools/ablate_libs/<name>.so (same per-translation-unit flags as the
# shipped build, plus the given ones).
#   tools/ablate.sh name1="-DADN_CF2=16 -DADN_RS2=4" name2="-DADN_ABLATE=3" ...
# Macros (adanerf_amd/csrc/tuning.hpp part 2; an override without -DADN_EXPERIMENT, which this script adds, is a compile error):
# ADN_ABLATE / ADN_ABLATE_S / ADN_ABLATE_G / ADN_ABLATE_DMA_BYTES (timing ablations, wrong results), ADN_CF2 / ADN_RS2 / ADN_NR2 (shading ring geometry,
# fragment registers), ADN_CF_S / ADN_RS_S / ADN_NR_S (split sampling kernel).  Closed experiments are plain constants in part 1 of that header.
# Run on the GPU box with tools/run_variants.sh
cd "$(dirname "$0")/.."
mkdir -p tools/ablate_libs
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  python -m adanerf_amd.build --out tools/ablate_libs/$name.so --flags="-DADN_EXPERIMENT $flags" > /dev/null &
done
wait
ls tools/ablate_libs
