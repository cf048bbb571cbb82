#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 PMC passes written by tools/collect_profiles.sh, plus the HBM
traffic estimate per launch with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled).

The guide leaves WRITE_SIZE uncalibrated ("calibrate on a known byte count in your own access pattern").  Three kernels
of this path store an exactly known byte count, and WRITE_SIZE reproduces each to the KiB, so it is used as reported:
shade_mlp16_kernel, one float4 per sample (config 2: 4 539 214 x 16 B = 70 925 KiB, reported 70 925.2; config 3 dense:
81 920 000 x 16 B = 1 280 000 KiB, reported 1 280 000.0); composite_kernel, 12 + 4 B per ray (640 000 rays = 10 000 KiB,
reported 10 000.0)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
names = {"shade_mlp16x2": "shade_mlp16x2_kernel", "shade_mlp16_gen_staged": "shade_mlp16_gen_staged_kernel", "sample_mlp16x3_gen": "sample_mlp16x3_gen_kernel",
         "shade_mlp16_gen": "shade_mlp16_gen_kernel", "shade_mlp32_gen": "shade_mlp32_gen_kernel",
         "sample_mlp_gen": "sample_mlp_gen_kernel", "refine_list": "refine_list_kernel",
         "shade_mlp16": "shade_mlp16_kernel", "shade_mlp32": "shade_mlp32_kernel", "sample_mlp16x3": "sample_mlp16x3_kernel",
         "sample_mlp16_kernel": "sample_mlp16_kernel", "sample_mlp_kernel": "sample_mlp_kernel", "select_kernel": "select_kernel",
         "select_rows": "select_rows_kernel", "expand_kernel": "expand_kernel", "scan_blocks": "scan_blocks_kernel",
         "composite_kernel": "composite_kernel", "composite_wave": "composite_wave_kernel", "dense_expand": "dense_expand_kernel"}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "pmc_*", "pmc_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = next((v for p, v in names.items() if p in row["Kernel_Name"]), None)
        if k is None:
            continue
        vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
out = {}
for k, d in vals.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    rec = {"counters": m, "launches_sampled": max(len(v) for v in d.values()), "mean_duration_ns_under_pmc": sum(dur[k]) / len(dur[k])}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        rec["hbm_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
        rec["hbm_bytes_note"] = ("(2 x FETCH_SIZE + WRITE_SIZE) KiB: gfx950 FETCH_SIZE counts 64 B per 128-B request; WRITE_SIZE checked against the "
                                 "exactly known stores of shade_mlp16_kernel / composite_kernel (tools/summarize_pmc.py)")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
        rec["kernel_cycles"] = cyc
        rec["mfma_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc if cyc else None
        rec["effective_clock_ghz"] = cyc / rec["mean_duration_ns_under_pmc"]
    out[k] = rec
try:
    from adanerf_amd.build import source_hash
    out["_meta"] = {"source_hash": source_hash(), "workload": os.environ.get("PMC_WORKLOAD", ""), "bench_args": os.environ.get("BENCH_ARGS", "")}
except Exception as e:      # noqa
    out["_meta"] = {"source_hash": None, "error": str(e)}
json.dump(out, sys.stdout, indent=1)
