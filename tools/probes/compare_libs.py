"""Bitwise comparison of two builds of the library (ADANERF_LIB_A / ADANERF_LIB_B, default: the in-tree build) on the raw
sampling-network outputs and on a whole frame -- for kernel variants that must not change a single bit."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import adanerf_oracle as O
from conftest import case_weights, load_case
import adanerf_amd
from adanerf_amd import build as B

libs = [os.environ.get("ADANERF_LIB_A") or B.library_path(), os.environ.get("ADANERF_LIB_B") or B.library_path()]
ok = True
for case, (w, h) in (("classroom_n8_thr02", (320, 200)), ("ndc_synthetic_n8", (256, 144)), ("barbershop_n4_thr015", (200, 120))):
    z, meta, sc = load_case(case)
    wts = case_weights(meta)
    td = tempfile.mkdtemp()
    O.write_model_dir(td, sc, wts)
    res = []
    for lib in libs:
        with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), precision="bf16", lib_path=lib) as r:
            r.set_camera(z["pose"], z["rot"])
            orc = r.empty((w * h, 128), np.float32)
            r.sample_mlp(0, w * h, orc, None)
            rgb, rgba, st = r.render_numpy()
            res.append((orc.numpy(), rgb, int(st.total_samples)))
    same_orc = np.array_equal(res[0][0], res[1][0])
    same_rgb = np.array_equal(res[0][1], res[1][1])
    print("%s: oracle values identical %s (max diff %.3g), frame identical %s, samples %d / %d" %
          (case, same_orc, float(np.abs(res[0][0] - res[1][0]).max()), same_rgb, res[0][2], res[1][2]))
    ok &= same_orc and same_rgb
sys.exit(0 if ok else 1)
