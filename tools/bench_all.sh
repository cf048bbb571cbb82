#!/bin/bash
# on the GPU box: one bench line per BASELINE configuration that fits one GPU -> gpurun_out/bench_all/<name>.json
cd "$(dirname "$0")/.."
O=gpurun_out/${ROUND:-r06}_bench_all; mkdir -p $O
# GUARDED=1 run ...: the line also carries guarded_mode (the opt-in guarded selection measured beside the headline, and what --sampling auto picks)
run() { n=$1; shift; g=--no-guarded-mode; [ "${GUARDED:-0}" = "1" ] && g=; python bench.py --no-speed-mode --no-exact-mode $g --no-split-mode "$@" 2>/dev/null | tail -1 > $O/$n.json; python - "$O/$n.json" "$n" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
q = d.get("quality") or {}
g = d.get("guarded_mode") or {}
box = (d.get("roofline") or {}).get("box") or {}
print("%-22s %7.1f FPS  spp %.2f  stages %s  frac %.3f  psnr %s%s  sclk %s MHz %s W" % (sys.argv[2], d["value"], d["config"]["mean_samples_per_ray"],
      {k: round(v, 3) for k, v in d["stage_ms_per_frame"].items()}, d["roofline"]["frac"], round(q.get("psnr_vs_oracle_db", 0), 1) if q else "-",
      ("  guarded %.1f FPS (%+.1f %%, auto: %s)" % (g["value"], 100 * g["ahead_of_the_headline"], g["auto_choice"])) if g else "",
      round(box["sclk_mhz_mean"]) if box.get("sclk_mhz_mean") else "?", round(box["power_w_mean"]) if box.get("power_w_mean") else "?"))
PY
}
GUARDED=1 run config2_bf16 --steps 30 --no-cpu-baseline
run config2_bf16_guarded --steps 30 --sampling guarded --no-cpu-baseline
run config2_bf16_guarded_audit_off --steps 30 --sampling guarded --guard-audit-period -1 --no-cpu-baseline
run config2_fp16 --steps 30 --precision fp16 --no-cpu-baseline
run config2_fp32 --steps 10 --precision fp32 --no-cpu-baseline
run config2_fp32sampling --steps 20 --sampling fp32 --no-cpu-baseline
GUARDED=1 run config4_thr01 --steps 30 --workload config4 --no-cpu-baseline
run config3_dense --steps 5 --warmup 2 --workload config3_dense --no-cpu-baseline
for t in 0.05 0.1 0.2 0.3 0.4; do GUARDED=1 run config5_ndc_fp16_thr$t --steps 20 --workload config5_ndc --precision fp16 --threshold $t --no-cpu-baseline; done
run config2_orbit16 --steps 32 --orbit 16 --no-cpu-baseline
run nerf_coarse_fine --steps 5 --warmup 2 --workload nerf_coarse_fine --no-cpu-baseline
run config2_fp16_sampling_only --steps 30 --sampling fp16 --no-cpu-baseline
run config5_ndc_fp16_thr0.2_guarded --steps 20 --workload config5_ndc --precision fp16 --threshold 0.2 --sampling guarded --no-cpu-baseline
run generic_6x128_bf16 --steps 10 --workload generic_6x128 --no-cpu-baseline
run generic_6x128_fp32 --steps 5 --workload generic_6x128 --precision fp32 --no-cpu-baseline
run generic_4x64_bf16 --steps 20 --workload generic_4x64 --no-cpu-baseline
run generic_5x256_bf16 --steps 20 --workload generic_5x256 --no-cpu-baseline
run generic_6x128_bf16_fp32sampling --steps 20 --workload generic_6x128 --sampling fp32 --no-cpu-baseline
