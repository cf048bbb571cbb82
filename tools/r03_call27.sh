#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
(timeout 300 python tools/probes/dense_shard_repro.py 40 24 3 8 60) > $O/r03_dense_shard_repro.log 2>&1; cat $O/r03_dense_shard_repro.log | cut -c1-300 | tail -40
