"""Debug probe: run-time-shaped split-precision sampling kernel against the oracle on a topology fixture; prints how the difference is
structured (constant per output bin = a bias block of the last layer, per ray = earlier layers)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import adanerf_oracle as O
from conftest import case_weights, load_case
import adanerf_amd
name = sys.argv[1] if len(sys.argv) > 1 else "syn_6x128_skip2"
z, meta, sc = load_case(name); wts = case_weights(meta)
d = tempfile.mkdtemp(); O.write_model_dir(d, sc, wts)
w, h = 64, 32
for smp in ("split", "fp32"):
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16", sampling=smp) as r:
        r.set_camera(z["pose"], z["rot"])
        buf = r.empty((w * h, 128), np.float32); rays = r.empty((w * h, 8), np.float32)
        r.sample_mlp(0, w * h, buf, rays)
        orc = buf.numpy(); rr = rays.numpy()
    nds, p = O.world_rays(O.generate_ray_directions(w, h, sc.fov), z["pose"], z["rot"], sc)
    ref = O.sampling_mlp(O.oracle_features(nds, p, sc), wts.net0)
    df = orc - ref
    print(smp, "max|d| %.3e  per-bin mean |mean_d| max %.3e  std over rays of d (mean over bins) %.3e  rays exact: p %.1e" %
          (np.abs(df).max(), np.abs(df.mean(0)).max(), df.std(0).mean(), np.abs(rr[:, :3] - p).max()))
    print("   first ray d[:8]", df[0, :8], " bin-mean d[:8]", df.mean(0)[:8])
