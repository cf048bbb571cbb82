#!/bin/bash
# Builds experiment variants of the library into tools/ablate_libs/<name>.so (same per-translation-unit flags as the
# shipped build, plus the given ones).
#   tools/ablate.sh name1="-DADN_CF=8 -DADN_RS=8" name2="-DADN_ABLATE=3" ...
# Macros (adanerf_amd/csrc/tuning.hpp; honoured only together with -DADN_EXPERIMENT, which this script adds):
# ADN_ABLATE / ADN_ABLATE_S (timing ablations, wrong results), ADN_CF / ADN_RS / ADN_NR (shade ring geometry, fragment
# registers), ADN_CF_S / ADN_RS_S / ADN_NR_S (sampling kernel), ADN_STAGGER, ADN_DMA_GRP, ADN_PAD, ADN_SEL_RPB, ADN_HANDSCHED / ADN_HANDSCHED_S (0: compiler-scheduled layers).
# Run on the GPU box with tools/run_variants.sh
cd "$(dirname "$0")/.."
mkdir -p tools/ablate_libs
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  python -m adanerf_amd.build --out tools/ablate_libs/$name.so --flags="-DADN_EXPERIMENT $flags" > /dev/null &
done
wait
ls tools/ablate_libs
