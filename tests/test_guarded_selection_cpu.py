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


@pytest.mark.parametrize("n_max,thr", [(8, 0.2), (4, 0.15), (16, 0.15), (1, 0.3)])
def test_pair_bound_rule(n_max, thr):
    """Untransformed outputs with a measured bound on (kept - candidate) difference errors (eps_pair < 2 eps): more rays are decided
    than under 2 eps, and a decided ray keeps its selection under every perturbation that respects BOTH bounds -- here the worst
    admissible one for a pair bound: a common offset of up to eps - eps_pair / 2 on the whole row (fully correlated errors) plus
    up to eps_pair / 2 of anything on top, so single values move by <= eps and differences by <= eps_pair."""
    rng = np.random.default_rng(77 + n_max)
    R = 4000
    y = (rng.standard_normal((R, 128)) * 0.05).astype(F32)
    for r in range(R):
        k = rng.integers(0, 20)
        y[r, rng.integers(0, 128, k)] = rng.uniform(0, 1.2, k)
        if r % 5 == 0:
            y[r, rng.integers(0, 128, 3)] = thr + rng.uniform(-0.03, 0.03, 3)
    eps, ep = 0.01, 0.008
    und2 = O.guard_undecided(y, n_max, thr, eps)
    und = O.guard_undecided(y, n_max, thr, eps, eps_pair=ep)
    assert (und <= und2).all() and und.sum() < und2.sum()      # never more conservative than 2 eps, and it buys something
    assert (O.guard_undecided(y, n_max, thr, eps, eps_pair=2 * eps) == und2).all()
    assert (O.guard_undecided(y, n_max, thr, eps, eps_pair=5 * eps) == und2).all()     # a bound above 2 eps is not a bound
    c0, b0, _ = O.select_adaptive(y, n_max, thr)
    e = np.full(R, F32(ep / 2))
    for common in (-1.0, 0.0, 1.0):
        for x in _adversarial(y, e, F32(thr), n_max, rng):
            x = (x + F32(common * (eps - ep / 2) * 0.999)).astype(F32)
            assert np.abs(x - y).max() <= eps
            assert O.guard_pair_error(y, x, n_max, thr, eps).max() <= ep * 1.0001
            c, b, _ = O.select_adaptive(x, n_max, thr)
            same = (c == c0) & (b == b0).all(axis=1)
            assert same[~und].all(), "a decided ray changed its selection under a perturbation inside both bounds"
    # ... and the pair bound is what carries it: with independent errors of the full eps the narrower rule would be wrong somewhere
    e = np.full(R, F32(eps))
    broke = False
    for x in _adversarial(y, e, F32(thr), n_max, rng):
        c, b, _ = O.select_adaptive(x, n_max, thr)
        broke |= bool((~((c == c0) & (b == b0).all(axis=1)))[~und].any())
    assert broke or n_max == 1


def test_pair_error_statistic():
    """guard_pair_error on rows built by hand: only (kept, candidate-not-kept) pairs count, with the sign that closes the gap."""
    n_max, thr, eps = 2, F32(0.2), F32(0.01)
    y = np.full((4, 128), -1.0, dtype=F32)
    x = y.copy()
    y[:, 3], y[:, 9], y[:, 20] = 0.9, 0.5, 0.49          # kept: bins 3, 9; candidate: bin 20 (within 2 eps of the cut 0.5)
    x[:] = y
    x[0, 9] -= 0.004; x[0, 20] += 0.003                  # kept over-read by 4e-3, candidate under-read by 3e-3: pair error 7e-3
    x[1, 9] += 0.004; x[1, 20] -= 0.003                  # the other sign opens the gap: 0
    x[2, 3] -= 0.006                                     # the larger kept value counts too: 6e-3 against the candidate's 0
    y[3, 20] = 0.47; x[3] = y[3]; x[3, 9] -= 0.009       # no candidate within 2 eps of the cut: 0
    got = O.guard_pair_error(y, x, n_max, thr, eps)
    assert np.allclose(got, [0.007, 0.0, 0.006, 0.0], atol=1e-6), got


@pytest.mark.parametrize("period", [1, 2, 4, 8, 16, 32])
def test_audit_rotation_covers_every_ray_once_per_period(period):
    """Over `period` consecutive frames every ray of every segment is audited exactly once; at a fixed phase 32 / period rays of a
    segment are, and neighbouring segments audit different lanes (no fixed image column)."""
    for seg in (0, 1, 5, 31, 32, 1000):
        seen = 0
        for phase in range(period):
            b = O.guard_audit_bits(period, phase, seg)
            assert bin(b).count("1") == 32 // period and (seen & b) == 0
            for j in range(32):
                assert ((b >> j) & 1) == (1 if ((j - phase - seg) & (period - 1)) == 0 else 0)
            seen |= b
        assert seen == 0xFFFFFFFF
    if period > 1:
        assert O.guard_audit_bits(period, 0, 0) != O.guard_audit_bits(period, 0, 1)
    assert O.guard_audit_bits(0, 3, 7) == 0


def test_audit_fill_never_adds_a_round_and_starves_no_ray():
    """ADANERF_FLAG_GUARD_AUDIT_FILL (restated in guard_refine_list): the list is every undecided ray plus a window of the frame's audit
    candidates that exactly fills the last round of the refinement pass (or all candidates when they fit) -- one more round only where
    that room is below a quarter of the candidates, so a frame never audits less than that; the window moves on by its own length from
    cycle to cycle, so over ceil(candidates / room) <= 4 cycles of one phase every candidate has been audited."""
    rng = np.random.default_rng(3)
    R, period, cap = 20000, 16, 4096
    extra_rounds = 0
    for frac in (0.02, 0.19, 0.21, 0.39, 0.405, 0.41, 0.97):
        und = rng.random(R) < frac
        n_und = int(und.sum())
        rounds = max(1, -(-n_und // cap))
        for phase in (0, 5, 15):
            full, full_a = O.guard_refine_list(und, period, phase)
            cand = full[full_a]
            assert np.array_equal(full[~full_a], np.flatnonzero(und)) and (np.diff(full) > 0).all()
            seen = np.zeros(R, bool)
            room = rounds * cap - n_und
            more = room < (cand.size + 3) // 4
            extra_rounds += more
            room = min(cand.size, room + (cap if more else 0))
            cycles = 1 if room >= cand.size else -(-cand.size // room)
            assert cycles <= 4 and room >= min(cand.size, (cand.size + 3) // 4)
            for cycle in range(cycles):
                rays, a = O.guard_refine_list(und, period, phase, cap, cycle)
                assert (np.diff(rays) > 0).all() and np.array_equal(rays[~a], np.flatnonzero(und))
                assert a.sum() == room and rays.size <= (rounds + more) * cap and np.isin(rays[a], cand).all()
                assert rays.size == (rounds + more) * cap or room == cand.size          # the last round is full, or every candidate is in
                seen[rays[a]] = True
            assert seen[cand].all(), "audit candidates left out for good"
    assert 0 < extra_rounds < 21            # the population has both kinds of frames
    # nothing undecided: one round's worth of audit still runs
    rays, a = O.guard_refine_list(np.zeros(R, bool), period, 3, cap, 0)
    assert a.all() and rays.size == min(cap, R // period)
