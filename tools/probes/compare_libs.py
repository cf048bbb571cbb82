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
from conftest import TOPOLOGY_CASES
from adanerf_amd import renderer as R
for case, (w, h) in [("classroom_n8_thr02", (320, 200)), ("ndc_synthetic_n8", (256, 144)), ("barbershop_n4_thr015", (200, 120)), ("classroom_dense128", (48, 32))] + \
        [(c, (160, 96)) for c in TOPOLOGY_CASES]:
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
            raw = r.buffer(R.BUF_RAW, np.float32, (int(st.total_samples), 4)) if r.info.batch_rays >= w * h else np.zeros((0, 4), np.float32)
            res.append((orc.numpy(), rgb, int(st.total_samples), raw))
    same_orc = np.array_equal(res[0][0], res[1][0])
    same_rgb = np.array_equal(res[0][1], res[1][1])
    same_raw = res[0][3].shape == res[1][3].shape and np.array_equal(res[0][3], res[1][3])
    print("%s: oracle values identical %s (max diff %.3g), frame identical %s, raw shading outputs identical %s (%d x 4, max |diff| %.3g), samples %d / %d" %
          (case, same_orc, float(np.abs(res[0][0] - res[1][0]).max()), same_rgb, same_raw, res[0][3].shape[0],
           float(np.abs(res[0][3] - res[1][3]).max()) if res[0][3].shape == res[1][3].shape and res[0][3].size else -1.0, res[0][2], res[1][2]))
    ok &= same_orc and same_rgb and same_raw
sys.exit(0 if ok else 1)
