#!/bin/bash
# Builds experiment variants of the library into tools/ablate_libs/<name>.so.
#   tools/ablate.sh name1="-DADN_CF=8 -DADN_RS=8" name2="-DADN_ABLATE=3" ...
# Macros: ADN_ABLATE / ADN_ABLATE_S (timing ablations, wrong results), ADN_CF / ADN_RS (shade ring
# geometry), ADN_CF_S / ADN_RS_S (sampling ring geometry).  Run on the GPU box with
#   for f in tools/ablate_libs/*.so; do ADANERF_LIB=$PWD/$f python bench.py --no-cpu-baseline ...; done
cd "$(dirname "$0")/.."
mkdir -p tools/ablate_libs
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $flags \
    adanerf_amd/csrc/adanerf_hip.hip adanerf_amd/csrc/format.cpp adanerf_amd/csrc/pack.cpp -o tools/ablate_libs/$name.so &
done
wait
ls tools/ablate_libs
