#!/bin/bash
# session 31: fuzz over all kinds on the current sources
cd /root/repo
O=gpurun_out/r04_s31; mkdir -p $O
FUZZ_ROUND2=1 FUZZ_ROUND3=1 timeout 900 python tests/fuzz_parity.py 150 9101 > $O/fuzz_150_seed9101.log 2>&1; tail -4 $O/fuzz_150_seed9101.log; grep -c " ok" $O/fuzz_150_seed9101.log; grep -v " ok" $O/fuzz_150_seed9101.log | grep "^case" | head
