#!/bin/bash
# rebuilds the library with -Rpass-analysis=kernel-resource-usage and prints the register/spill report
# for kernels matching $1 (regex)
cd "$(dirname "$0")/.."
python -m adanerf_amd.build --out /tmp/adanerf_report.so --flags="-Rpass-analysis=kernel-resource-usage" 2>&1 \
  | grep -E "error|Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size" | grep -A8 -E "${1:-.}|error" | sed 's/.*remark: //; s/ \[-Rpass.*//'
