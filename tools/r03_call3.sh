#!/bin/bash
# GPU session 3 of round 3: timing ablations of the two-blocks-per-wave shading kernel, SQ counters base vs sb2, full suite with
# the tightened floors.
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out
for v in base sb2 sb2bw sb2cf32 sb2cf32bw sb2nr4 sb2_a1 sb2_a2 sb2_a3 sb2_a4 sb2_a8 sb2_a15 sb2_a16 sb2_a32; do
  f=tools/ablate_libs/$v.so
  ADANERF_LIB=$R/$f timeout 120 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-speed-mode 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['value'],1), {k: round(x,3) for k,x in r['stage_ms_per_frame'].items()})"
done > $O/r03_ablate_sb2.log 2>&1
cat $O/r03_ablate_sb2.log
ADANERF_LIB=$R/tools/ablate_libs/sb2.so timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "shade_mlp_matches_oracle or frame_low_precision_psnr or render_is_deterministic or full_size" 2>&1 | tail -2
export TMPDIR=/tmp
for v in base sb2; do
  OUT=$O/r03_pmc_$v; mkdir -p $OUT
  for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    N=$(echo $P | cut -d" " -f1)
    (cd /tmp && ADANERF_LIB=$R/tools/ablate_libs/$v.so timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_$N -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-speed-mode > $OUT/pmc_$N.log 2>&1)
  done
  PMC_WORKLOAD=config2 python tools/summarize_pmc.py $OUT > $O/r03_pmc_summary_$v.json
  find $OUT -name "*.csv" -size +2M -delete
done
python - <<'PY'
import json
for v in ("base", "sb2"):
    d = json.load(open("gpurun_out/r03_pmc_summary_%s.json" % v))
    k = d.get("shade_mlp16_kernel", {})
    print(v, {c: round(x) for c, x in k.get("counters", {}).items()}, k.get("mfma_pipe_busy_frac"), k.get("effective_clock_ghz"), k.get("mean_duration_ns_under_pmc"))
PY
export ADANERF_MEASURED_LOG=$O/r03_measured3.log; rm -f $ADANERF_MEASURED_LOG
python -m pytest tests -m gpu -q > $O/r03_pytest_all3.log 2>&1; tail -6 $O/r03_pytest_all3.log
