"""Repro of fuzz seed 5301 case 60 (flaky): barbershop weights, 40 x 24, dense (N = 128, thr = 0), bf16, 8 contexts on one GPU with
8-row strips (ranks 3..7 own no rows).  Renders the unsharded frame and the sharded + assembled frame repeatedly and reports where
they differ."""
import dataclasses, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import adanerf_amd
import adanerf_oracle as O
from conftest import load_case, case_weights

z, meta, sc = load_case("barbershop_n4_thr015"); wts = case_weights(meta)
sc = dataclasses.replace(sc, num_samples=128, threshold=0.0)
w, h = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 24
world, rows = int(sys.argv[3]) if len(sys.argv) > 3 else 8, int(sys.argv[4]) if len(sys.argv) > 4 else 8
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
d = tempfile.mkdtemp()
O.write_model_dir(d, sc, wts)
pose = np.array(sc.view_cell_center, np.float32); rot = O.camera_rotation(30.0, 5.0)
with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16") as r:
    r.set_camera(pose, rot)
    frames = [r.render_numpy()[1].copy() for _ in range(reps)]
base = frames[0]
print("unsharded: %d of %d repeats differ from the first" % (sum(not np.array_equal(f, base) for f in frames[1:]), reps - 1))
rs = [adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16", shard_rank=k, shard_world=world, strip_rows=rows) for k in range(world)]
for q in rs:
    q.init(); q.set_camera(pose, rot)
root = rs[0]
stride = root.info.rays_local_max * 4
gathered = root.empty((world, max(root.info.rays_local_max, 1), 4), np.uint8)
image = root.empty((w * h, 4), np.uint8)
pay = [None] + [q.empty((max(q.info.rays_local_max, 1), 4), np.uint8) for q in rs[1:]]
from adanerf_amd import renderer as R, sharding
with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16") as r:
    r.set_camera(pose, rot)
    r.render_numpy()
    raw_full = r.buffer(R.BUF_RAW, np.float32, (w * h * 128, 4)).reshape(w * h, 128, 4)
    orc_full = r.buffer(R.BUF_ORACLE, np.float32, (w * h, 128))
bad = 0
for it in range(reps):
    root.render(gathered.ptr, None)
    for k in range(1, world):
        rs[k].render(pay[k], None)
        root.gather_from(gathered.ptr + k * stride, rs[k], pay[k], stride)
    root.assemble_strips(gathered, image)
    root.sync()
    img = image.numpy()
    if not np.array_equal(img, base):
        bad += 1
        for k in range(world):      # which sample / component of which rank's raw buffer differs from the unsharded run?
            n = rs[k].info.rays_local
            if not n:
                continue
            pix = sharding.local_to_pixel(w, h, rows, world, k)
            raw_k = rs[k].buffer(R.BUF_RAW, np.float32, (n * 128, 4)).reshape(n, 128, 4)
            orc_k = rs[k].buffer(R.BUF_ORACLE, np.float32, (n, 128))
            dr = np.argwhere(raw_k != raw_full[pix])
            do = np.argwhere(orc_k != orc_full[pix])
            for (ray, b, comp) in dr[:6]:
                sidx = int(ray) * 128 + int(b)
                print("   rank %d raw differs: local ray %d bin %d comp %d  sample index %d (tile %d, in-tile %d: wave %d, block %d, lane %d)  %r vs %r" %
                      (k, ray, b, comp, sidx, sidx // 256, sidx % 256, (sidx % 256) // 64, ((sidx % 256) // 32) % 2, sidx % 32, raw_k[ray, b, comp], raw_full[pix][ray, b, comp]))
            if len(do):
                print("   rank %d oracle differs at %d entries, first %s" % (k, len(do), do[0]))
        px = np.nonzero((img != base).any(axis=1))[0]
        print("repeat %d: %d pixels differ; rows %s; first: pixel %d sharded %s unsharded %s" %
              (it, len(px), sorted(set((px // w).tolist())), px[0], img[px[0]], base[px[0]]))
print("sharded: %d of %d repeats differ from the unsharded frame; rays_local per rank %s" % (bad, reps, [q.info.rays_local for q in rs]))
for q in rs:
    q.close()
