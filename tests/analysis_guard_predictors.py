"""CPU-only analysis behind profiles/r03_guarded_selection.md ("what else was tried for the band"): emulates the plain-fp16 sampling
engine (weights and activations rounded to fp16, fp32 accumulate) against exact fp32 on 60 000 rays of the config-2 frame and asks
  (1) does any cheap per-ray quantity predict a ray's error well enough for a per-ray band?
  (2) how do error and refined fraction move when only some layers run in plain fp16?
Test infrastructure (imports the oracle); not collected by pytest.    python tests/analysis_guard_predictors.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import adanerf_oracle as O                      # noqa: E402
from conftest import case_weights, load_case    # noqa: E402


def f16(a):
    return a.astype(np.float16).astype(np.float32)


def main():
    z, meta, sc = load_case("classroom_n8_thr02")
    wts = case_weights(meta)
    w, h = 800, 800
    dirs = O.generate_ray_directions(w, h, sc.fov).reshape(-1, 3)
    pose = np.array(sc.view_cell_center, np.float32)
    rot = O.camera_rotation(100.0, 0.0)
    idx = np.random.default_rng(0).choice(w * h, 60000, replace=False)
    nds, p = O.world_rays(dirs[idx], pose, rot, sc)
    x = O.oracle_features(nds, p, sc)
    net = wts.net0
    n = len([k for k in net if k.endswith(".weight")])

    def run(plain):      # plain: set of layers evaluated with fp16 operands
        hq, norms = x.copy(), []
        for i in range(n):
            W, b = net["layers.%d.weight" % i], net["layers.%d.bias" % i]
            hq = (f16(hq) @ f16(W).T + b) if i in plain else (hq @ W.T + b)
            if i + 1 < n:
                hq = np.maximum(hq, 0)
                norms.append((np.abs(hq).max(1), np.sqrt((hq * hq).sum(1))))
        return hq, norms

    he, _ = run(set())
    hq, norms = run(set(range(n)))
    err = np.abs(hq - he).max(1)
    print("plain fp16 vs exact: max %.3e  median %.3e  99 %% %.3e" % (err.max(), np.median(err), np.quantile(err, 0.99)))

    def refined(out, eps):      # per-ray eps, quantised to 1/8 octave
        und = np.zeros(len(out), bool)
        q = np.round(np.log2(np.maximum(eps, 1e-6)) * 8) / 8
        for v in np.unique(q):
            m = q == v
            und[m] = O.guard_undecided(out[m], sc.num_samples, sc.threshold, float(2 ** v))
        return float(und.mean())

    print("global band 2 x max = %.3e: %.1f %% of the rays undecided" % (2 * err.max(), 100 * refined(hq, np.full(len(err), 2 * err.max()))))
    print("(1) per-ray predictors g: band_ray = 2 x max(err / g) x g")
    for name, g in (("largest |output|", np.abs(hq).max(1)), ("last hidden layer, Linf", norms[-1][0]), ("last hidden layer, L2", norms[-1][1]),
                    ("sum of the L2 norms of all hidden layers", sum(nm[1] for nm in norms))):
        r = err / np.maximum(g, 1e-9)
        print("  %-42s corr(err, g) %+.2f   max / median of err / g %5.1f   undecided %.1f %%" %
              (name, np.corrcoef(err, g)[0, 1], r.max() / np.median(r), 100 * refined(hq, 2 * r.max() * g)))
    print("(2) the LAST k layers in plain fp16, the first 8 - k exact (3 MFMAs per term)")
    for k in range(1, n + 1):
        out, _ = run(set(range(n - k, n)))
        e = np.abs(out - he).max(1)
        und = refined(out, np.full(len(e), 2 * e.max()))
        first = (k + 3 * (n - k)) / n
        print("  k = %d: max err %.2e  undecided %.1f %%   MFMAs per term: first pass %.2f + refinement %.2f = %.2f" %
              (k, e.max(), 100 * und, first, 3 * und, first + 3 * und))
    print("    the FIRST k layers in plain fp16, the rest exact")
    for k in range(1, n):
        out, _ = run(set(range(k)))
        e = np.abs(out - he).max(1)
        print("  k = %d: max err %.2e  undecided %.1f %%" % (k, e.max(), 100 * refined(out, np.full(len(e), 2 * e.max()))))


if __name__ == "__main__":
    main()
