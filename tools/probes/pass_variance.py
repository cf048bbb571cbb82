"""Frame-rate of six 30-frame passes per sampling mode (a 5 s pause before the fourth): how much a short side measurement of bench.py can be off."""
import sys, time, tempfile, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as Bn, adanerf_amd
from adanerf_amd import modeldir as M
w, h, n, thr, tag = Bn.WORKLOADS["config2"]
td = tempfile.mkdtemp(); scene, _ = Bn.build_model_dir(td, tag, n, thr)
pose = np.array(scene["view_cell_center"], np.float32); rot = M.camera_rotation(100.0, 0.0)
for smp in ("split", "guarded", "split", "guarded"):
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), precision="bf16", sampling=smp) as r:
        r.set_camera(pose, rot); out = r.empty((w * h, 4), np.uint8)
        for _ in range(5): r.render(out, None)
        r.sync(); res = []
        for p in range(6):
            if p == 3: time.sleep(5.0)
            t = time.perf_counter()
            for _ in range(30): r.render(out, None)
            r.sync(); res.append(30 / (time.perf_counter() - t))
        print(smp, ["%.1f" % x for x in res], flush=True)
