"""how much shading work sits behind an (almost) opaque front?  For every sample of a frame the transmittance
|T| in front of it (src/nerf_raymarch_common.py:123-135: alpha = sigmoid(raw[3]) * oracle value, T *= 1 - alpha + 1e-10, samples in ascending
depth) from the frame's own buffers (ADANERF_BUF_RAW, _SAMPLE_W, _RAY_COUNTS / _RAY_OFFSETS; fp32 shading so that the histogram is the model's, not a
16-bit path's).  A sample behind |T| < eps changes the pixel by < alpha_max * eps per sample (alpha <= 1.78 for oracle values <= 1.78: measured
below), so an opt-in "shading cut-off" mode could skip it.  Prints, per workload / pose: samples per ray, and the fraction of samples whose front
transmittance is below 2^-6 ... 2^-14, split by position in the ray (index 0..3 / 4+): the second part is what a two-pass shading (samples 0-3, then
the rest of the rays still transparent) could save.
usage: python tools/probes/transmittance.py [config2|config4|config5_ndc] [n_orbit_poses]"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import adanerf_amd
import bench as Bn
from adanerf_amd import modeldir as M

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
n_orbit = int(sys.argv[2]) if len(sys.argv) > 2 else 16
w, h, n_max, thr, tag = Bn.WORKLOADS[wl]
td = tempfile.mkdtemp()
scene, _ = Bn.build_model_dir(td, tag, n_max, thr)
c = np.array(scene["view_cell_center"], np.float32)
size = np.array(scene["view_cell_size"], np.float32)
poses = [("centre, yaw 100", c, M.camera_rotation(100.0, 0.0) if tag != "ndc_random_init" else np.eye(3, dtype=np.float32))]
for i in range(n_orbit):
    th = 2.0 * np.pi * i / n_orbit
    p = c + 0.3 * size * np.array([np.cos(th), np.sin(th), 0.25 * np.sin(2 * th)], np.float32)
    poses.append(("orbit %d" % i, p.astype(np.float32), M.camera_rotation(100.0 + 360.0 * i / n_orbit, 0.0) if tag != "ndc_random_init" else np.eye(3, dtype=np.float32)))
eps_list = [2.0 ** -k for k in (6, 8, 10, 12, 14)]
tot = {"samples": 0, "rays": 0, "behind": np.zeros(len(eps_list)), "behind_tail": np.zeros(len(eps_list)), "tail": 0, "amax": 0.0}
rows = []
with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), precision="fp32", sampling="split") as r:
    for name, p, rot in poses:
        r.set_camera(p, rot)
        st = r.render(None, r.empty((w * h, 3), np.float32), stats=True) if False else None
        out = r.empty((w * h, 4), np.uint8)
        st = r.render(out, None, stats=True)
        S, R = int(st.total_samples), w * h
        raw = r.buffer(6, np.float32, (S, 4))
        sw = r.buffer(5, np.float32, (S,))
        cnt = r.buffer(3, np.int32, (R,))
        off = r.buffer(2, np.int32, (R,))
        assert int(cnt.sum()) == S and np.array_equal(off[1:], np.cumsum(cnt)[:-1])
        alpha = (1.0 / (1.0 + np.exp(-raw[:, 3].astype(np.float64)))) * sw.astype(np.float64)
        # front transmittance of sample k of its ray = prod_{j<k} (1 - alpha_j + 1e-10): segmented exclusive cumulative product, in log space with signs
        f = 1.0 - alpha + 1e-10
        idx = np.arange(S) - np.repeat(off, cnt)                       # position of the sample in its ray
        logf = np.log(np.maximum(np.abs(f), 1e-300))
        cs = np.cumsum(logf)
        start = np.repeat(cs[off] - logf[off], cnt)                    # cumulative sum in front of the ray's first sample
        front = np.exp(cs - logf - start)                              # |T| in front of each sample
        row = {"pose": name, "samples_per_ray": S / R, "alpha_max": float(alpha.max()), "alpha_min": float(alpha.min())}
        for e, eps in enumerate(eps_list):
            b = front < eps
            row["behind_2^-%d" % int(-np.log2(eps))] = float(b.mean())
            tot["behind"][e] += b.sum()
            tot["behind_tail"][e] += (b & (idx >= 4)).sum()
        tot["samples"] += S
        tot["rays"] += R
        tot["tail"] += int((idx >= 4).sum())
        tot["amax"] = max(tot["amax"], float(alpha.max()))
        rows.append(row)
        print(json.dumps(row))
print("== %s: %d poses, %.2f samples per ray, alpha in [.., %.3f]" % (wl, len(poses), tot["samples"] / tot["rays"], tot["amax"]))
for e, eps in enumerate(eps_list):
    print("   front |T| < 2^-%-2d : %6.3f %% of all samples   (%6.3f %% of all samples are such samples at position >= 4 of their ray; %5.1f %% of the samples at position >= 4)"
          % (int(-np.log2(eps)), 100 * tot["behind"][e] / tot["samples"], 100 * tot["behind_tail"][e] / tot["samples"], 100 * tot["behind_tail"][e] / max(tot["tail"], 1)))
