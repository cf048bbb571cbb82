O=$PWD/gpurun_out/r04_s18; mkdir -p $O
for wl in generic_6x128 generic_5x256 generic_4x64; do
  echo "== $wl"; BENCH_ARGS="--no-speed-mode --no-exact-mode --no-sustained-probe --workload $wl" STEPS=10 bash tools/run_variants.sh
done 2>&1 | tee $O/variants_generic_ablate_bias.log
