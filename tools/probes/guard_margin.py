"""How much head-room the calibrated guard band has over many poses: config-2 model at 800 x 800, P random poses inside the view cell.
Per pose: the monitor's running maxima (largest |fp16 - split| over all 128 outputs of a re-evaluated ray so far, largest pair error),
violations, audited rays and audit mismatches, refined rays; for every 8th pose also the true maximum over ALL raw outputs of the frame
(both engines through the stage API)."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import adanerf_amd
import bench as Bn
from adanerf_amd import modeldir as M

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h, n_max, thr, tag = Bn.WORKLOADS["config2"]
td = tempfile.mkdtemp()
scene, _ = Bn.build_model_dir(td, tag, n_max, thr)
c, size = np.array(scene["view_cell_center"], np.float32), np.array(scene["view_cell_size"], np.float32)
rng = np.random.default_rng(5)
with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), precision="bf16", sampling="guarded") as rg, \
        adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), sampling="split") as rs, \
        adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), sampling="fp16") as rf:
    bs, bf = rs.empty((w * h, 128), np.float32), rf.empty((w * h, 128), np.float32)
    worst = 0.0
    for i in range(P):
        pose = (c + rng.uniform(-0.45, 0.45, 3).astype(np.float32) * size).astype(np.float32)
        rot = M.camera_rotation(float(rng.uniform(0, 360)), float(rng.uniform(-35, 35)))
        rg.set_camera(pose, rot)
        st = rg.render(None, None, stats=True)
        rg.lib.adanerf_get_info(rg.handle, rg.info)
        rec = dict(pose=i, eps=float(rg.info.guard_eps), eps_pair=float(rg.info.guard_eps_pair), monitor_running_max=float(st.guard_max_seen),
                   monitor_pair_running_max=float(st.guard_pair_seen), violations=int(st.guard_violations), audited=int(st.guard_audited),
                   audit_mismatches=int(st.guard_audit_mismatch), widened=int(st.guard_widened),
                   refined_frac=st.rays_refined / float(w * h), spp=st.total_samples / float(w * h))
        if i % 8 == 0:
            for r, b in ((rs, bs), (rf, bf)):
                r.set_camera(pose, rot)
                r.sample_mlp(0, w * h, b, None)
            d = float(np.abs(bs.numpy() - bf.numpy()).max())
            worst = max(worst, d)
            rec["true_max_all_outputs"] = d
        print(json.dumps(rec), flush=True)
    print(json.dumps(dict(poses=P, eps=rec["eps"], eps_pair=rec["eps_pair"], monitor_max=rec["monitor_running_max"], monitor_pair_max=rec["monitor_pair_running_max"],
                          violations=rec["violations"], audited=rec["audited"], audit_mismatches=rec["audit_mismatches"], widened=rec["widened"],
                          worst_true_max_sampled=worst)))
