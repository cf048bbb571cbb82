#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python tools/probes/dense_stage_isolation.py 60 > gpurun_out/r03_dense_stage_isolation.log 2>&1; cut -c1-300 gpurun_out/r03_dense_stage_isolation.log | tail -30
