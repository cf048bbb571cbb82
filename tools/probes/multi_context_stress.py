"""Frames of the adaptive (fused-selection) path rendered by several contexts on one GPU with their launches interleaved: every frame
of a context must be byte-identical to its first (the integer wave exchanges of the selection / scan kernels next to other kernels;
companion of dense_stage_isolation.py).  args: width height contexts frames sampling"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import adanerf_amd
import adanerf_oracle as O
from conftest import load_case, case_weights
w, h = int(sys.argv[1]), int(sys.argv[2])
world, M, smp = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
z, meta, sc = load_case("classroom_n8_thr02"); wts = case_weights(meta)
d = tempfile.mkdtemp(); O.write_model_dir(d, sc, wts)
rs = [adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16", sampling=smp, shard_rank=k, shard_world=world, strip_rows=5) for k in range(world)]
for q in rs:
    q.init(); q.set_camera(z["pose"], z["rot"])
first = []
for q in rs:
    b = q.empty((q.info.rays_local, 4), np.uint8); q.render(b, None); q.sync(); first.append(b.numpy().copy())
bad, total = 0, 0
CH = 20
for c0 in range(0, M, CH):
    bufs = [[q.empty((q.info.rays_local, 4), np.uint8) for _ in range(CH)] for q in rs]
    for i in range(CH):
        for q, b in zip(rs, bufs):
            q.render(b[i], None)
    for q in rs: q.sync()
    for k, bb in enumerate(bufs):
        for b in bb:
            x = b.numpy(); total += 1
            if not np.array_equal(x, first[k]):
                bad += 1
                px = np.nonzero((x != first[k]).any(axis=1))[0]
                print("context %d: %d pixels differ, first %d: %s vs %s" % (k, len(px), px[0], x[px[0]], first[k][px[0]]))
        for b in bb: b.free()
print("%s sampling, %d contexts x %d frames of %d x %d (%d rays each): %d of %d frames differ from the context's first" % (smp, world, M, w, h, rs[0].info.rays_local, bad, total))
for q in rs: q.close()
