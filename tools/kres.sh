#!/bin/bash
# one line per kernel matching $1: registers, scratch, spills, LDS (build with -Rpass-analysis=kernel-resource-usage into /tmp)
cd "$(dirname "$0")/.."
python -m adanerf_amd.build --out /tmp/adanerf_report.so --flags="-Rpass-analysis=kernel-resource-usage ${FLAGS:-}" 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|Spill|ScratchSize|LDS Size" | sed 's/.*remark: //; s/ \[-Rpass.*//' | awk '/error/{print} /Function Name/{name=$3} /VGPRs:/{v=$2} /AGPRs/{a=$2} /ScratchSize/{sc=$3} /SGPRs Spill/{ss=$3} /VGPRs Spill/{vs=$3} /LDS Size/{print name, "V="v, "A="a, "scratch="sc, "sspill="ss, "vspill="vs, "lds="$4}' | grep -E "error|${1:-.}" | c++filt
