"""Per-rank share cost on ONE GPU: renders rank r's strips of the config-2 frame for world = 1, 2, 4, 8 and prints the
stage times, so the 8-GPU frame time (max over ranks, before the gather) can be projected without an 8-GPU node."""
import json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import adanerf_amd
import bench as Bn
from adanerf_amd import modeldir as M

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
w, h, n_max, thr, tag = Bn.WORKLOADS[wl]
if len(sys.argv) > 2:
    thr = float(sys.argv[2])
worlds = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (1, 2, 4, 8)
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
sampling = sys.argv[5] if len(sys.argv) > 5 else "split"      # what bench.py measures by default (round 5)
from adanerf_amd import sharding
td = tempfile.mkdtemp()
scene, _ = Bn.build_model_dir(td, tag, n_max, thr)
pose = np.array(scene["view_cell_center"], np.float32)
rot = M.camera_rotation(100.0, 0.0) if tag != "ndc_random_init" else np.eye(3, dtype=np.float32)
rows = []
for world in worlds:
    for rank in range(world):
        with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), precision=prec, sampling=sampling, shard_rank=rank, shard_world=world,
                                       strip_rows=sharding.balanced_strip_rows(h, world)) as r:
            r.set_camera(pose, rot)
            out = r.empty((r.info.rays_local_max, 4), np.uint8)
            for _ in range(5):
                r.render(out, None)
            r.sync()
            t0 = time.perf_counter()
            K = 40
            for _ in range(K):
                r.render(out, None)
            r.sync()
            wall = (time.perf_counter() - t0) / K * 1e3
            r.set_profiling(True)
            for _ in range(10):
                r.render(out, None)
            st, frames = r.collect_stats()
            r.set_profiling(False)
            rec = dict(world=world, rank=rank, rays=r.info.rays_local, samples=st.total_samples / frames, wall_ms=wall,
                       sample_ms=st.ms_sample_mlp / frames, compact_ms=st.ms_compact / frames,
                       shade_ms=st.ms_shade_mlp / frames, composite_ms=st.ms_composite / frames,
                       rays_refined=st.rays_refined / frames, audited_total=int(st.guard_audited))
            rows.append(rec)
            print(json.dumps(rec), flush=True)
base = rows[0]["wall_ms"] if worlds[0] == 1 else float("nan")
for world in [x for x in worlds if x > 1]:
    ws = [x for x in rows if x["world"] == world]
    mx = max(x["wall_ms"] for x in ws)
    sm = [x["samples"] for x in ws]
    print("world %d: max share %.3f ms -> projected %.0f FPS before the gather, efficiency %.1f%%, sample imbalance max/mean %.3f"
          % (world, mx, 1e3 / mx, 100 * base / (world * mx), max(sm) / (sum(sm) / len(sm))))
