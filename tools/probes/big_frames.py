import sys, os, tempfile, numpy as np
sys.path.insert(0, os.getcwd())
import bench as Bn, adanerf_amd
from adanerf_amd import modeldir as M
td = tempfile.mkdtemp(); scene, _ = Bn.build_model_dir(td, "sample_pavillon_16", 8, 0.2)
for (w, h) in ((3840, 2160), (7680, 4320), (801, 599), (1, 1), (33, 1)):
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h)) as r:
        r.set_camera(np.array(scene["view_cell_center"], np.float32), M.camera_rotation(100.0, 0.0))
        out = r.empty((w * h, 4), np.uint8)
        st = r.render(out, None, stats=True)
        st = r.render(out, None, stats=True)
        a = out.numpy()
        print(w, h, "ms %.2f samples/ray %.3f" % (st.ms_total, st.total_samples / (w * h)), "alpha ok", bool((a[:, 3] == 255).all()), "mean rgb", a[:, :3].mean(axis=0).round(1), "overflow", st.sampling_overflow)
