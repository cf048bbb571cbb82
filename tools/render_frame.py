#!/usr/bin/env python3
"""Renders one frame of the bench workload's model (the reference's exported classroom weights fixture) and
writes it as PNG -- a visual sanity artifact for profiles/.   python tools/render_frame.py out.png [w h yaw pitch]"""
import json, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import adanerf_amd
from adanerf_amd import modeldir as M
from adanerf_amd.png import write_png
out = sys.argv[1]
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (400, 400)
yaw, pitch = (float(sys.argv[4]), float(sys.argv[5])) if len(sys.argv) > 5 else (100.0, 0.0)
gold = os.path.join(ROOT, "tests", "golden")
z = np.load(os.path.join(gold, "weights_sample_pavillon_16.npz"))
s = json.load(open(os.path.join(gold, "scene_sample_pavillon_16.json")))
td = tempfile.mkdtemp()
M.write_model_dir(td, dict(view_cell_center=s["view_cell_center"], view_cell_size=s["view_cell_size"], depth_range=s["depth_range"],
                           fov=s["fov"], max_depth=s["max_depth"], num_samples=8, threshold=0.2),
                  {k[3:]: z[k] for k in z.files if k.startswith("n0/")}, {k[3:]: z[k] for k in z.files if k.startswith("n1/")})
with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), precision="bf16") as r:
    r.set_camera(np.array(s["view_cell_center"], np.float32), M.camera_rotation(yaw, pitch))
    rgb, rgba, st = r.render_numpy()
write_png(out, rgba[:, :3].reshape(h, w, 3))
print("wrote %s: %dx%d, %.2f samples/ray, %.3f ms" % (out, w, h, st.total_samples / float(w * h), st.ms_total))
