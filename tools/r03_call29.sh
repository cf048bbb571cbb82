#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 600 python tools/probes/dense_stage_isolation.py 60 > $O/r03_dense_stage_isolation_dpp.log 2>&1; grep -v "fp32 composite" $O/r03_dense_stage_isolation_dpp.log | cut -c1-250 | tail -12; grep -c "fp32 composite" $O/r03_dense_stage_isolation_dpp.log
timeout 300 python tools/probes/dense_shard_repro.py 40 24 3 8 100 2>&1 | tail -3 | cut -c1-200
timeout 900 python -m pytest tests -x -q -m gpu -k "dense or pdf or coarse or classic or aux or depth or disp or transform or softmax or config3" 2>&1 | tail -4
