#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_final8; mkdir -p $O
ADANERF_MEASURED_LOG=$PWD/$O/parity_measured.log timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
