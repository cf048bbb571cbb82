#!/bin/bash
cd "$(dirname "$0")/.."
(timeout 300 python tools/probes/multi_context_stress.py 400 320 3 200 guarded; timeout 300 python tools/probes/multi_context_stress.py 400 320 4 200 split; timeout 300 python tools/probes/multi_context_stress.py 800 800 2 100 guarded) 2>&1 | tail -12 | cut -c1-220 | tee gpurun_out/r03_multi_context_stress.log
