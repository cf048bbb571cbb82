"""Guarded two-precision selection (ADANERF_SAMPLING_GUARDED): what the band has to cover and what it costs.
For each workload: the distribution of |plain-fp16 output - split-precision output| over every raw output of the frame
(the quantity guard_eps must bound), then per eps the fraction of rays the rule sends to the refinement pass, the rays
whose fp16 selection differs from the split engine's and are NOT caught (must be 0), and the measured stage times.
  python tools/probes/guard_band.py [config2 config4 config5_ndc ...]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import numpy as np
import adanerf_oracle as O, adanerf_amd
import bench as Bn
from adanerf_amd import modeldir as M

EPS = [float(e) for e in os.environ.get('GUARD_EPS_LIST', '0 0.002 0.005 0.01').split()]      # 0: the band the library calibrates for the model
for wl in (sys.argv[1:] or ["config2", "config4", "config5_ndc"]):
    thrs = [None] if wl != "config5_ndc" else [0.05, 0.2, 0.4]
    for thr_o in thrs:
        w, h, n_max, thr, tag = Bn.WORKLOADS[wl]
        if thr_o is not None:
            thr = thr_o
        td = tempfile.mkdtemp()
        scene, _ = Bn.build_model_dir(td, tag, n_max, thr)
        pose = np.array(scene["view_cell_center"], dtype=np.float32)
        rot = M.camera_rotation(100.0, 0.0) if tag != "ndc_random_init" else np.eye(3, dtype=np.float32)
        orc = {}
        for smp in ("split", "fp16"):
            with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), sampling=smp) as r:
                r.set_camera(pose, rot)
                buf = r.empty((w * h, 128), np.float32)
                r.sample_mlp(0, w * h, buf, None)
                orc[smp] = buf.numpy()
        diff = np.abs(orc["fp16"] - orc["split"])
        per_ray = diff.max(axis=1)
        print(json.dumps({"workload": wl, "thr": thr, "rays": w * h, "max_abs_diff": float(diff.max()),
                          "quantiles_per_ray_max": {q: float(np.quantile(per_ray, q)) for q in (0.5, 0.9, 0.99, 0.999, 0.9999)},
                          "max_abs_value": float(np.abs(orc["split"]).max())}))
        sub = slice(0, None, 4)
        xs, ys = orc["split"][sub], orc["fp16"][sub]
        cs, bs, _ = O.select_adaptive(xs, n_max, thr)
        cy, by, _ = O.select_adaptive(ys, n_max, thr)
        differs = (cs != cy) | (bs != by).any(axis=1)
        rel = per_ray[sub]
        for eps in EPS:
            with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), sampling="guarded", guard_eps=eps) as r:
                r.set_camera(pose, rot)
                for _ in range(3):
                    r.render(None, None)
                ts = [r.render(None, None, stats=True) for _ in range(5)]
                r.lib.adanerf_get_info(r.handle, r.info)
                row = {"workload": wl, "thr": thr, "eps_requested": eps, "eps": float(r.info.guard_eps)}
                row.update(sample_ms=float(np.median([t.ms_sample_mlp for t in ts])), total_ms=float(np.median([t.ms_total for t in ts])),
                           rays_refined=int(ts[0].rays_refined), samples=int(ts[0].total_samples),
                           monitor_max_seen=float(ts[-1].guard_max_seen), monitor_violations=int(ts[-1].guard_violations))
            und = O.guard_undecided(ys, n_max, thr, row["eps"])
            row.update({"undecided_frac": float(und.mean()), "fp16_differs_frac": float(differs.mean()),
                        "differs_and_not_caught": int((differs & ~und).sum()), "rays_whose_error_exceeds_eps": int((rel > row["eps"]).sum())})
            print(json.dumps(row))
        for smp in ("split", "fp16"):
            with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), sampling=smp) as r:
                r.set_camera(pose, rot)
                for _ in range(3):
                    r.render(None, None)
                ts = [r.render(None, None, stats=True) for _ in range(5)]
                print(json.dumps({"workload": wl, "thr": thr, "sampling": smp, "sample_ms": float(np.median([t.ms_sample_mlp for t in ts])),
                                  "total_ms": float(np.median([t.ms_total for t in ts])), "samples": int(ts[0].total_samples)}))
