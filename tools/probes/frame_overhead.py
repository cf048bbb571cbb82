"""Host + launch overhead per frame: renders a tiny frame many times (GPU work ~0) with and without per-frame sync."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import adanerf_amd
from adanerf_amd import modeldir as M
td = tempfile.mkdtemp()
n0, n1 = M.random_init_weights(0)
M.write_model_dir(td, dict(view_cell_center=(0.783, -3.19, 1.39), view_cell_size=(0.7, 0.7, 0.2), depth_range=(0.154, 8.358),
                           fov=1.1386, max_depth=8.798, num_samples=8, threshold=0.2), n0, n1)
for (w, h) in [(32, 32), (800, 100)]:
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h)) as r:
        r.set_camera(np.array([0.783, -3.19, 1.39], np.float32), M.camera_rotation(100, 0))
        out = r.empty((w * h, 4), np.uint8)
        for _ in range(20):
            r.render(out, None)
        r.sync()
        t0 = time.perf_counter()
        N = 500
        for _ in range(N):
            r.render(out, None)
        t1 = time.perf_counter()
        r.sync()
        t2 = time.perf_counter()
        print("%dx%d: enqueue %.1f us/frame, end-to-end %.1f us/frame" % (w, h, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
