#!/bin/bash
# session 33: MFMA probe with the shading kernel's operand delivery (LDS fragments, DMA refill, conversions)
cd /root/repo
O=gpurun_out/r04_s33; mkdir -p $O
adanerf_amd/bin/mfma_peak > $O/mfma_peak_dataflow.log 2>&1; cat $O/mfma_peak_dataflow.log | tail -16
