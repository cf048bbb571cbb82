"""What two frames in flight buy a small share: rank 0's strips of the config-2 frame for world = 8 (and the whole frame, world = 1)
rendered K times by ONE context on one stream, and alternately by TWO contexts on two streams (bench.py --frames-in-flight 2)."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import adanerf_amd
import bench as Bn
from adanerf_amd import modeldir as M, sharding

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
w, h, n_max, thr, tag = Bn.WORKLOADS[wl]
td = tempfile.mkdtemp()
scene, _ = Bn.build_model_dir(td, tag, n_max, thr)
pose = np.array(scene["view_cell_center"], np.float32)
rot = M.camera_rotation(100.0, 0.0) if tag != "ndc_random_init" else np.eye(3, dtype=np.float32)
dev = torch.device("cuda", 0)
for world in (8, 4, 1):
    res = {}
    for fif in (1, 2):
        rs, streams, outs = [], [], []
        for i in range(fif):
            r = adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), precision="bf16", sampling="guarded", shard_rank=0, shard_world=world,
                                           strip_rows=sharding.balanced_strip_rows(h, world))
            r.init(); r.set_camera(pose, rot)
            s = torch.cuda.Stream(device=dev); r.set_stream(s.cuda_stream)
            rs.append(r); streams.append(s); outs.append(torch.zeros((r.info.rays_local_max, 4), dtype=torch.uint8, device=dev))
        K = 200 if world > 1 else 60
        for k in range(10):
            rs[k % fif].render(outs[k % fif], None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            rs[k % fif].render(outs[k % fif], None)
        torch.cuda.synchronize()
        res[fif] = (time.perf_counter() - t0) / K * 1e3
        for r in rs:
            r.close()
    print(json.dumps({"workload": wl, "world": world, "share_ms_one_frame_at_a_time": res[1], "share_ms_two_in_flight": res[2], "gain": res[1] / res[2]}), flush=True)
