#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "survives" 2>&1 | tail -25
