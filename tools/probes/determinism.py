#!/usr/bin/env python3
"""Same launch twice (and against a second context): are the sampling-net outputs and the shading-net outputs bit-identical,
and if not, which rays / samples (wave of the workgroup, tile) differ?  Used while hand-scheduling the MLP layers."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import adanerf_amd
from adanerf_amd import renderer as R
from conftest import load_case, case_weights
from adanerf_amd import modeldir

z, meta, sc = load_case("classroom_n8_thr02")
wts = case_weights(meta)
d = tempfile.mkdtemp()
import adanerf_oracle as O
O.write_model_dir(d, sc, wts)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (400, 300)
for sampling in ("split", "fp16"):
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16", sampling=sampling, keep_oracle=True) as r:
        r.set_camera(z["pose"], z["rot"])
        n = w * h
        outs = []
        for rep in range(3):
            o = r.empty((n, 128), np.float32)
            r.sample_mlp(0, n, o, None)
            outs.append(o.numpy().copy())
        ref = outs[0]
        for k, o in enumerate(outs[1:]):
            bad = np.flatnonzero((o != ref).any(axis=1))
            tile = 128 if sampling == "split" else 256
            print(sampling, "sample_mlp run", k + 1, "rays differing:", bad.size, "of", n,
                  "waves:", np.bincount((bad % tile) // 32, minlength=tile // 32).tolist() if bad.size else "-",
                  "max abs diff:", float(np.abs(o - ref).max()))
        # whole frames
        a = r.render_numpy(); b = r.render_numpy()
        print(sampling, "render twice: rgb equal", np.array_equal(a[0], b[0]), "samples", a[2].total_samples, b[2].total_samples)
        # shading stage on the frame's own samples, three times
        tot = int(r.buffer(R.BUF_TOTAL, np.int32, (1,))[0])
        raws = []
        for rep in range(3):
            r.render_numpy()
            raws.append(r.buffer(R.BUF_RAW, np.float32, (tot, 4)).copy())
        for k, o in enumerate(raws[1:]):
            bad = np.flatnonzero((o != raws[0]).any(axis=1))
            print(sampling, "shade raw run", k + 1, "samples differing:", bad.size, "of", tot,
                  "waves:", np.bincount((bad % 256) // 32, minlength=8).tolist() if bad.size else "-",
                  "max abs diff:", float(np.abs(o - raws[0]).max()) if bad.size else 0.0)
