"""Cost of adanerf_compact by selection branch: 640 000 random rows, thresholds chosen so that (a) every ray has
<= N candidates (ballots only), (b) every ray has > N (bisection), (c) none (arg-max), plus the classroom oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import adanerf_amd, bench as Bn
import tempfile
td = tempfile.mkdtemp()
scene, _ = Bn.build_model_dir(td, "sample_pavillon_16", 8, 0.2)
n = 640000
rng = np.random.default_rng(0)
orc = rng.random((n, 128), dtype=np.float32)
with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, 800, 800)) as r:
    d = r.to_device(orc)
    off = r.empty((n,), np.int32); cnt = r.empty((n,), np.int32)
    key = r.empty((n * 8,), np.uint32); sw = r.empty((n * 8,), np.float32); tot = r.empty((1,), np.int32)
    for name, thr in (("<=N candidates (thr 0.97)", 0.97), (">N candidates (thr 0.5)", 0.5), (">N, tight (thr 0.9)", 0.9), ("none (thr 2)", 2.0)):
        for _ in range(3):
            r.compact(d, n, 8, thr, off, cnt, key, sw, tot)
        r.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            r.compact(d, n, 8, thr, off, cnt, key, sw, tot)
        r.sync()
        print("%-28s %.1f us  mean count %.2f" % (name, (time.perf_counter() - t0) / 20 * 1e6, cnt.numpy().mean()))
