#!/bin/bash
# rebuilds the library with -Rpass-analysis=kernel-resource-usage and prints the register/spill report
# for kernels matching $1 (regex)
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Rpass-analysis=kernel-resource-usage \
  adanerf_amd/csrc/adanerf_hip.hip adanerf_amd/csrc/format.cpp adanerf_amd/csrc/pack.cpp -o adanerf_amd/lib/libadanerf_hip.so 2>&1 \
  | grep -E "error|Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size" | grep -A8 -E "${1:-.}|error" | sed 's/.*remark: //; s/ \[-Rpass.*//'
