"""CPU-only analysis behind profiles/r04_lab_log.md section 10 ("a band per output rank / sort-position-specific eps"):
emulates the plain-fp16 sampling engine against an fp64 evaluation on 4 x 30 000 rays of the classroom model and asks whether the error
of an output depends on its VALUE (so that the outputs near the threshold, or at the N-th / (N+1)-th rank, could get a narrower band
than the global worst case).  It does not: the maximum is 2.6e-3 ... 3.4e-3 in every value range from 0 to 1, and a band proportional to
a0 + |v| is wider near the threshold than the global one.  Test infrastructure (imports the oracle); not collected by pytest.
    python tests/analysis_guard_value_band.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import adanerf_oracle as O
from conftest import case_weights, load_case
def f16(a): return a.astype(np.float16).astype(np.float32)
z, meta, sc = load_case("classroom_n8_thr02")
wts = case_weights(meta)
w, h = 800, 800
dirs = O.generate_ray_directions(w, h, sc.fov).reshape(-1, 3)
pose = np.array(sc.view_cell_center, np.float32)
rng = np.random.default_rng(0)
net = wts.net0
n = len([k for k in net if k.endswith(".weight")])
def run(x, plain):
    hq = x.copy()
    for i in range(n):
        W, b = net["layers.%d.weight" % i], net["layers.%d.bias" % i]
        hq = (f16(hq) @ f16(W).T + b) if plain else (hq.astype(np.float64) @ W.T.astype(np.float64) + b).astype(np.float32)
        if i + 1 < n: hq = np.maximum(hq, 0)
    return hq
outs=[]
for yaw in (100.0, 10.0, 200.0, 290.0):
    rot = O.camera_rotation(yaw, 0.0)
    idx = rng.choice(w * h, 30000, replace=False)
    nds, p = O.world_rays(dirs[idx], pose, rot, sc)
    x = O.oracle_features(nds, p, sc)
    outs.append((run(x, False), run(x, True)))
he = np.concatenate([o[0] for o in outs]); hq = np.concatenate([o[1] for o in outs])
err = np.abs(hq - he)
print("rays", he.shape[0], "global max err %.3e" % err.max(), "value range", he.min(), he.max())
edges = [-10, -1, -0.5, -0.2, -0.1, 0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.4, 0.6, 1.0, 2.0, 10]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (he >= lo) & (he < hi)
    if m.sum(): print("value [%5.2f,%5.2f): n %9d  max err %.3e  99.99%% %.3e  median %.3e" % (lo, hi, m.sum(), err[m].max(), np.quantile(err[m], 0.9999), np.median(err[m])))
# relative model: err <= a + b*|v| : find max of err/(a0+|v|)
for a0 in (0.05, 0.1, 0.2, 0.5):
    r = err / (a0 + np.abs(he)); print("a0 %.2f: max err/(a0+|v|) = %.3e -> band at v=0.2: %.3e (x2: %.3e), at v=1: %.3e" % (a0, r.max(), r.max()*(a0+0.2), 2*r.max()*(a0+0.2), r.max()*(a0+1)))
# the undecided fraction with a value-dependent band eps(v) = 2*rmax*(a0+|v|)
N, thr = sc.num_samples, sc.threshold
def undecided_valueband(y, epsf):
    # conservative restatement: a ray is decided iff (a) none of top-N within eps(v) of thr; (b) no value outside the kept set within [cut - 2eps] ... use per-value eps
    srt = np.sort(y, axis=1)[:, ::-1]
    e = epsf(srt)
    top = srt[:, :N]; etop = e[:, :N]
    a = (np.abs(top - thr) <= etop).any(1)
    cnt = (y >= thr).sum(1)
    vN = srt[:, N-1]; eN = e[:, N-1]
    # candidates for flipping at the cut: values below cut whose upper bound exceeds the cut's lower bound
    kept = np.minimum(cnt, N)
    und = a.copy()
    for k in range(srt.shape[0]):
        c = kept[k]
        if c == 0:
            # arg-max fallback: top1 vs top2
            if srt[k,0]-e[k,0] <= srt[k,1]+e[k,1]: und[k]=True
            continue
        if cnt[k] > N:  # top-N cut binds: v_N vs v_{N+1}
            if srt[k,N-1]-e[k,N-1] <= srt[k,N]+e[k,N]: und[k]=True
    return und
for a0 in (0.1, 0.2):
    rmax = (err / (a0 + np.abs(he))).max()
    und = undecided_valueband(hq, lambda v: 2*rmax*(a0+np.abs(v)))
    print("value band a0=%.2f: undecided %.1f %%" % (a0, 100*und.mean()))
und = undecided_valueband(hq, lambda v: np.full_like(v, 2*err.max()))
print("global band (same simplified rule): undecided %.1f %%" % (100*und.mean()))
print("oracle rule global band:", 100*O.guard_undecided(hq, N, thr, float(2*err.max())).mean())
