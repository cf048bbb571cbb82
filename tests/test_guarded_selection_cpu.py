"""The guard-band rule of the two-precision selection (ADANERF_SAMPLING_GUARDED), checked on the CPU with the oracle's
restatement of it: wherever the rule says "decided", the selection must not depend on which values within the band
the exact engine would have produced.  (The GPU tests check that the device code implements this rule and that the
whole two-pass path reproduces the split-precision engine's selections.)"""
import numpy as np
import pytest

import adanerf_oracle as O

F32 = np.float32


def _adversarial(y, e, thr, n_max, rng):
    """Perturbations of y inside the band that try to flip the selection: values near thr are pushed across it, the
    n_max-th / (n_max+1)-th largest are pushed towards each other, the rest is random."""
    outs = []
    for mode in range(4):
        d = rng.uniform(-1, 1, y.shape).astype(F32)
        if mode == 1:      # everything towards the threshold and past it
            d = np.sign(thr - y).astype(F32)
        elif mode == 2:    # the large ones down, the small ones up (closes gaps between ranks)
            srt = -np.sort(-y, axis=1)
            mid = 0.5 * (srt[:, n_max - 1] + srt[:, n_max])
            d = np.sign(mid[:, None] - y).astype(F32)
        elif mode == 3:    # the top one down, everyone else up (the arg-max fallback)
            d = np.ones_like(y)
            d[np.arange(len(y)), y.argmax(axis=1)] = -1
        outs.append((y + e[:, None] * d * F32(0.999)).astype(F32))
    return outs


@pytest.mark.parametrize("n_max,thr", [(8, 0.2), (4, 0.15), (16, 0.15), (1, 0.3), (7, 0.05), (8, 2.0)])
def test_decided_rays_cannot_flip(n_max, thr):
    rng = np.random.default_rng(n_max * 1000 + int(thr * 100))
    R = 4000
    # peaky rows like an oracle net's: a few bins well above the threshold, a sea of small values, a band population
    y = (rng.standard_normal((R, 128)) * 0.05).astype(F32)
    for r in range(R):
        k = rng.integers(0, 20)
        y[r, rng.integers(0, 128, k)] = rng.uniform(0, 1.2, k)
        if r % 5 == 0:
            y[r, rng.integers(0, 128, 3)] = thr + rng.uniform(-0.03, 0.03, 3)
    eps = 0.01
    und = O.guard_undecided(y, n_max, thr, eps)
    assert 0.02 < und.mean() < 0.9            # the test population exercises both outcomes
    c0, b0, _ = O.select_adaptive(y, n_max, thr)
    e = np.full(R, F32(eps))
    for x in _adversarial(y, e, F32(thr), n_max, rng):
        c, b, _ = O.select_adaptive(x, n_max, thr)
        same = (c == c0) & (b == b0).all(axis=1)
        assert same[~und].all(), "a decided ray changed its selection under a perturbation inside the band"


def test_rows_built_to_straddle_the_band():
    n_max, thr, eps = 8, F32(0.2), F32(0.01)
    base = np.full(128, -1.0, dtype=F32)

    def row(**kv):
        r = base.copy()
        for k, v in kv.items():
            r[int(k[1:])] = v
        return r

    rows = {
        "clear":                 (row(b3=0.9, b40=0.5, b77=0.3), False),
        "value_just_inside":     (row(b3=0.9, b40=0.2 + 0.0099), True),
        "value_just_below":      (row(b3=0.9, b40=0.2 - 0.0099), True),
        "value_just_outside_hi": (row(b3=0.9, b40=0.2 + 0.0101), False),
        "value_just_outside_lo": (row(b3=0.9, b40=0.2 - 0.0101), False),
        "argmax_close_runner":   (row(b3=0.1, b40=0.1 - 0.019), True),
        "argmax_clear_runner":   (row(b3=0.1, b40=0.1 - 0.021), False),
        "argmax_near_thr":       (row(b3=0.2 - 0.005), True),
        "nan":                   (row(b3=0.9, b5=np.nan), True),
        "inf":                   (row(b3=np.inf, b5=0.5), True),
    }
    # nine values above the threshold: the cut binds between the 8th and the 9th
    nine = {"b%d" % (10 * i): 0.9 - 0.05 * i for i in range(8)}
    rows["cut_close"] = (row(**nine, b100=0.9 - 0.05 * 7 - 0.019), True)
    rows["cut_clear"] = (row(**nine, b100=0.9 - 0.05 * 7 - 0.021), False)
    rows["cut_tie"] = (row(**nine, b100=0.9 - 0.05 * 7), True)
    y = np.stack([v[0] for v in rows.values()])
    exp = np.array([v[1] for v in rows.values()])
    got = O.guard_undecided(y, n_max, thr, eps)
    assert list(got) == list(exp), {k: (bool(g), bool(e)) for k, g, e in zip(rows, got, exp) if g != e}


@pytest.mark.parametrize("transform", ["sigmoid", "softmax"])
def test_transformed_values_band(transform):
    """The band is stated on the RAW outputs; through the sigmoid it shrinks by 4, through the softmax it becomes
    relative.  Property: perturb the raw values within eps, transform, select -> decided rays keep their selection."""
    rng = np.random.default_rng(5)
    R, n_max, eps = 3000, 8, 0.01
    raw = (rng.standard_normal((R, 128)) * (2.0 if transform == "sigmoid" else 1.5)).astype(F32)
    thr = 0.6 if transform == "sigmoid" else 0.012
    losses0 = "BCEWithLogitsLoss" if transform == "sigmoid" else "CrossEntropyLoss"
    y = O.oracle_transform(raw, losses0)
    und = O.guard_undecided(y, n_max, thr, eps, transform)
    assert 0.01 < und.mean() < 0.99
    c0, b0, _ = O.select_adaptive(y, n_max, thr)
    for _ in range(4):
        x = O.oracle_transform((raw + rng.uniform(-eps, eps, raw.shape) * 0.999).astype(F32), losses0)
        c, b, _ = O.select_adaptive(x, n_max, thr)
        same = (c == c0) & (b == b0).all(axis=1)
        assert same[~und].all()
