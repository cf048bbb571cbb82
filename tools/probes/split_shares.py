"""Intra-frame concurrency for small shares: every rank's share of the frame at world = W rendered by ONE context on one stream, and
as P sub-shares (virtual ranks r*P .. r*P + P - 1 of a world of W * P) by P contexts on P streams -- the same frame, no second frame
in flight, so a frame's latency does not grow (it shrinks).  Prints, per (W, P), the slowest rank's ms per frame and the projected
efficiency against the single-GPU frame (what bench.py --gpus W does per rank with --sub-shares P)."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import adanerf_amd
import bench as Bn
from adanerf_amd import modeldir as M, sharding

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
worlds = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 8, 4, 2)
parts_list = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (1, 2, 3, 4)
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
sampling = sys.argv[5] if len(sys.argv) > 5 else "split"      # what bench.py measures by default (round 5)
w, h, n_max, thr, tag = Bn.WORKLOADS[wl]
td = tempfile.mkdtemp()
scene, _ = Bn.build_model_dir(td, tag, n_max, thr)
pose = np.array(scene["view_cell_center"], np.float32)
rot = M.camera_rotation(100.0, 0.0) if tag != "ndc_random_init" else np.eye(3, dtype=np.float32)
dev = torch.device("cuda", 0)
base = None
for world in worlds:
    for parts in parts_list:
        vw = world * parts
        worst = None
        for rank in range(world):
            rs, streams, outs = [], [], []
            for k in range(parts):
                r = adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), precision=prec, sampling=sampling, shard_rank=rank * parts + k, shard_world=vw,
                                               strip_rows=sharding.balanced_strip_rows(h, vw))
                r.init(); r.set_camera(pose, rot)
                s = torch.cuda.Stream(device=dev); r.set_stream(s.cuda_stream)
                rs.append(r); streams.append(s); outs.append(torch.zeros((r.info.rays_local_max, 4), dtype=torch.uint8, device=dev))
            K = 200 if world > 1 else 60
            for _ in range(10):
                for r, o in zip(rs, outs):
                    r.render(o, None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                for r, o in zip(rs, outs):
                    r.render(o, None)
            torch.cuda.synchronize()
            thr_ms = (time.perf_counter() - t0) / K * 1e3
            # latency of ONE frame: all parts launched together on an idle GPU, host waits for the last
            lat = []
            for _ in range(20):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for r, o in zip(rs, outs):
                    r.render(o, None)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t0) * 1e3)
            rays = sum(r.info.rays_local for r in rs)
            for r in rs:
                r.close()
            rec = {"rank": rank, "rays": rays, "ms_per_frame": round(thr_ms, 4), "latency_ms_median": round(float(np.median(lat)), 4)}
            if worst is None or rec["ms_per_frame"] > worst["ms_per_frame"]:
                worst = rec
        if world == 1 and parts == 1:
            base = worst["ms_per_frame"]
        out = dict({"workload": wl, "world": world, "sub_shares": parts, "slowest": worst})
        if base:
            out["efficiency_vs_single_gpu_frame"] = round(base / (world * worst["ms_per_frame"]), 4)
        print(json.dumps(out), flush=True)
