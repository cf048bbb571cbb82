O=$PWD/gpurun_out/r04_s9; mkdir -p $O; R=$PWD; export ADANERF_MEASURED_LOG=$O/parity_measured.log
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -n 6 $O/pytest_gpu.log
( time FUZZ_ROUND2=1 FUZZ_ROUND3=1 timeout 900 python tests/fuzz_parity.py 150 7002 ) > $O/fuzz_150_seed7002.log 2>&1; tail -n 5 $O/fuzz_150_seed7002.log; grep "FAIL" $O/fuzz_150_seed7002.log | head -20
