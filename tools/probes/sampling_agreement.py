"""Selection agreement of the sampling engines with the fp32 reference on the golden crops: max |raw output error|,
fraction of rays with the identical bin set / identical count, for split-fp16 (default) and plain fp16 (speed mode)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import numpy as np
import adanerf_oracle as O, adanerf_amd
from conftest import load_case, case_weights
for name in ["classroom_n8_thr02", "classroom_n16_thr015", "barbershop_n4_thr015", "ndc_synthetic_n8", "synthetic_fixed8"]:
    z, meta, sc = load_case(name); wts = case_weights(meta)
    d = tempfile.mkdtemp(); O.write_model_dir(d, sc, wts)
    c = meta["crop"]; x0, y0, cw, ch = c[:4]; stride = c[4] if len(c) > 4 else 1
    w = meta["w"]
    idx = np.concatenate([(y0 + i * stride) * w + x0 + np.arange(cw) * stride for i in range(ch)])
    for smp in ("split", "fp16"):
        with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, meta["w"], meta["h"]), sampling=smp) as r:
            r.set_camera(z["pose"], z["rot"])
            buf = r.empty((meta["w"] * meta["h"], 128), np.float32)
            r.sample_mlp(0, meta["w"] * meta["h"], buf, None)
            orc = buf.numpy()[idx]
        cnt, bins, _ = O.select_adaptive(orc, sc.num_samples, sc.threshold)
        same = (cnt == z["sel_count"]) & (bins == z["sel_bins"]).all(axis=1)
        print("%-24s %-6s rays %5d max|err| %.2e identical bin sets %.4f identical counts %.4f" %
              (name, smp, len(idx), np.abs(orc - z["oracle_out"]).max(), same.mean(), (cnt == z["sel_count"]).mean()))
