# session helper (round 6): bench every variant in tools/ablate_libs REPS times, then a parity subset against the variants named in PARITY
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for rep in $(seq ${REPS:-2}); do STEPS=20 tools/run_variants.sh 2>&1; done | tee gpurun_out/r06/variants_$1.log
for v in $PARITY; do echo "== parity $v"; ADANERF_LIB=$PWD/tools/ablate_libs/$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "${PARITY_K:-sample_mlp_matches or fused_selection or full_size_frame or golden or bit_identical or shade_mlp_matches_oracle or psnr or select or sin_or_cos or ray_features}" 2>&1 | tail -3; done | tee gpurun_out/r06/variants_$1_parity.log
