"""Which stage of the dense frame is not reproducible when several contexts run on one GPU at the same time?  Three contexts (the three
strip shards of a 40 x 24 dense frame) each run ONE stage M times back to back through the stage API on fixed inputs, the contexts'
launches interleaved; every output is compared with the first."""
import ctypes as C, dataclasses, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import adanerf_amd
import adanerf_oracle as O
from adanerf_amd import renderer as R
from conftest import load_case, case_weights

z, meta, sc = load_case("barbershop_n4_thr015"); wts = case_weights(meta)
sc = dataclasses.replace(sc, num_samples=128, threshold=0.0)
w, h, world, rows, M = 40, 24, 3, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 40
d = tempfile.mkdtemp(); O.write_model_dir(d, sc, wts)
pose = np.array(sc.view_cell_center, np.float32); rot = O.camera_rotation(30.0, 5.0)
rs = [adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16", shard_rank=k, shard_world=world, strip_rows=rows) for k in range(world)]
def dev(q, which):
    p, nb = C.c_void_p(), C.c_size_t()
    q._check(q.lib.adanerf_get_buffer(q.handle, which, C.byref(p), C.byref(nb)))
    return p.value
st = []
for q in rs:
    q.init(); q.set_camera(pose, rot)
    out = q.empty((q.info.rays_local, 4), np.uint8)
    q.render(out, None); q.sync()
    n = q.info.rays_local
    keys = q.to_device(np.arange(n * 128, dtype=np.uint32))
    st.append(dict(n=n, raw=dev(q, R.BUF_RAW), w=dev(q, R.BUF_ORACLE), off=dev(q, R.BUF_RAY_OFFSETS), cnt=dev(q, R.BUF_RAY_COUNTS), rays=dev(q, R.BUF_RAYS),
                   total=dev(q, R.BUF_TOTAL), keys=keys, frame=out.numpy().copy()))
def report(name, outs):
    for k in range(world):
        ref = outs[k][0]
        bad = [i for i in range(1, M) if not np.array_equal(outs[k][i], ref)]
        msg = ""
        if bad:
            dd = np.argwhere(outs[k][bad[0]] != ref)
            msg = " first diff at %s: %r vs %r (%d entries)" % (dd[0].tolist(), outs[k][bad[0]][tuple(dd[0])], ref[tuple(dd[0])], len(dd))
        print("%s, context %d: %d of %d repeats differ from the first%s" % (name, k, len(bad), M - 1, msg))
# (a) compositing alone
bufs = [[q.empty((s["n"], 4), np.uint8) for _ in range(M)] for q, s in zip(rs, st)]
for i in range(M):
    for q, s, b in zip(rs, st, bufs):
        q.composite(s["raw"], s["w"], s["off"], s["cnt"], s["n"], None, b[i])
for q in rs: q.sync()
outs = [[b.numpy() for b in bb] for bb in bufs]
report("composite", outs)
print("   composite vs the frame's own pixels: %s" % [bool(np.array_equal(outs[k][0], st[k]["frame"])) for k in range(world)])
# (b) shading alone
bufs = [[q.empty((s["n"] * 128, 4), np.float32) for _ in range(M)] for q, s in zip(rs, st)]
for i in range(M):
    for q, s, b in zip(rs, st, bufs):
        q.shade_mlp(s["rays"], s["keys"], s["total"], s["n"] * 128, b[i])
for q in rs: q.sync()
report("shade_mlp", [[b.numpy() for b in bb] for bb in bufs])
# (c) sampling alone
bufs = [[q.empty((s["n"], 128), np.float32) for _ in range(M)] for q, s in zip(rs, st)]
for i in range(M):
    for q, s, b in zip(rs, st, bufs):
        q.sample_mlp(0, s["n"], b[i], None)
for q in rs: q.sync()
report("sample_mlp", [[b.numpy() for b in bb] for bb in bufs])
# (d) whole frames, contexts interleaved, each into its own buffer
bufs = [[q.empty((s["n"], 4), np.uint8) for _ in range(M)] for q, s in zip(rs, st)]
for i in range(M):
    for q, b in zip(rs, bufs):
        q.render(b[i], None)
for q in rs: q.sync()
report("render", [[b.numpy() for b in bb] for bb in bufs])
# (f) context 0 runs ONE stage repeatedly on fixed inputs while contexts 1 and 2 render whole frames next to it
junk = [[q.empty((s["n"], 4), np.uint8) for _ in range(2)] for q, s in zip(rs, st)]
def with_neighbours(name, launch, make, reads):
    bufs = [make() for _ in range(M)]
    for i in range(M):
        launch(bufs[i])
        for k in (1, 2):
            rs[k].render(junk[k][i & 1], None)
    for q in rs: q.sync()
    o = [reads(b) for b in bufs]
    bad = [i for i in range(1, M) if not np.array_equal(o[i], o[0])]
    msg = ""
    if bad:
        dd = np.argwhere(o[bad[0]] != o[0])
        msg = " first diff at %s: %r vs %r (%d entries)" % (dd[0].tolist(), o[bad[0]][tuple(dd[0])], o[0][tuple(dd[0])], len(dd))
    print("%s on context 0 next to whole frames of contexts 1, 2: %d of %d repeats differ%s" % (name, len(bad), M - 1, msg))
q0, s0 = rs[0], st[0]
with_neighbours("composite", lambda b: q0.composite(s0["raw"], s0["w"], s0["off"], s0["cnt"], s0["n"], None, b), lambda: q0.empty((s0["n"], 4), np.uint8), lambda b: b.numpy())
# ... the same with the fp32 colours: which channels move, and by how much?
bufs = [q0.empty((s0["n"], 3), np.float32) for _ in range(4 * M)]
for i in range(4 * M):
    q0.composite(s0["raw"], s0["w"], s0["off"], s0["cnt"], s0["n"], bufs[i], None)
    for k in (1, 2):
        rs[k].render(junk[k][i & 1], None)
for q in rs: q.sync()
o = [b.numpy() for b in bufs]
ref = np.median(np.stack(o), axis=0)
for i, x in enumerate(o):
    dd = np.argwhere(x != ref)
    for (ray, ch) in dd[:4]:
        print("   fp32 composite repeat %d: ray %d channel %d: %.9g vs %.9g (diff %.3e); the ray's three channels %s vs %s" % (i, ray, ch, x[ray, ch], ref[ray, ch], x[ray, ch] - ref[ray, ch], x[ray], ref[ray]))
with_neighbours("shade_mlp", lambda b: q0.shade_mlp(s0["rays"], s0["keys"], s0["total"], s0["n"] * 128, b), lambda: q0.empty((s0["n"] * 128, 4), np.float32), lambda b: b.numpy())
with_neighbours("sample_mlp", lambda b: q0.sample_mlp(0, s0["n"], b, None), lambda: q0.empty((s0["n"], 128), np.float32), lambda b: b.numpy())
# (e) whole frames, ONE context at a time
for k, q in enumerate(rs):
    bb = [q.empty((st[k]["n"], 4), np.uint8) for _ in range(M)]
    for i in range(M):
        q.render(bb[i], None)
    q.sync()
    o = [b.numpy() for b in bb]
    print("render, context %d alone: %d of %d repeats differ" % (k, sum(not np.array_equal(x, o[0]) for x in o[1:]), M - 1))
for q in rs: q.close()
