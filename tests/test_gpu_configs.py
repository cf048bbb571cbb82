"""GPU tests at the sizes and settings of BASELINE.json's configurations 3, 4 and 5, the compositing multipliers other
than accumulationMult = alpha, and the RCCL branch of bench.py.  Same conventions as test_gpu_parity.py: the HIP path
through the C ABI against the oracle on the rows / crops the oracle finishes in seconds, size-independent properties at
full size.  Run with `pytest -m gpu` on an MI355X box."""
import dataclasses
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import adanerf_oracle as O
from conftest import AUX_CASES, BINS_CASES, COARSE_FINE_CASES, ENCODING_CASES, MULT_CASES, NORM_CASES, ROOT, TOPOLOGY_CASES, case_weights, check_identical, load_case, record, residual_budget

import adanerf_amd
from adanerf_amd import renderer as R
from adanerf_amd import sharding
from test_gpu_parity import golden_samples, model_dir, same_bin_sets, small_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    adanerf_amd.build_library()


@pytest.fixture(scope="module")
def classroom(tmp_path_factory):
    z, meta, sc = load_case("classroom_n8_thr02")
    wts = case_weights(meta)
    return z, meta, sc, wts


def _dir(tmp_path_factory, sc, wts, tag):
    return model_dir(tmp_path_factory, sc, wts, tag)


def render_sharded(d, w, h, pose, rot, world, strip_rows, **kw):
    """Every rank's strips rendered by its own context on this box's one GPU, then adanerf_assemble_strips."""
    parts, rmax, samples = [], None, []
    for rank in range(world):
        with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), shard_rank=rank, shard_world=world, strip_rows=strip_rows, **kw) as r:
            r.set_camera(pose, rot)
            _, rgba, st = r.render_numpy()
            rmax = r.info.rays_local_max
            assert r.info.rays_local == sharding.rays_local(w, h, strip_rows, world, rank)
            pad = np.zeros((rmax, 4), np.uint8)
            pad[:rgba.shape[0]] = rgba
            parts.append(pad)
            samples.append(int(st.total_samples))
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), shard_rank=0, shard_world=world, strip_rows=strip_rows, **kw) as r:
        img = r.empty((w * h, 4), np.uint8)
        r.assemble_strips(r.to_device(np.concatenate(parts)), img)
        r.sync()
        return img.numpy(), samples


def check_rows_against_oracle(sc, wts, w, h, pose, rot, rows, cnt, rgb, min_same, min_psnr, tag):
    for row in rows:
        ref = O.render_frame(sc, wts, w, h, pose, rot, rows=(row, row + 1))
        sl = slice(row * w, (row + 1) * w)
        same = cnt[sl] == ref["count"]
        p = O.psnr(rgb[sl][same], ref["rgb"][same])
        err = float(np.abs(rgb[sl][same] - ref["rgb"][same]).max())
        record(tag, row=row, identical_counts=float(same.mean()), psnr_db=p, max_abs=err)
        check_identical(same, tag, residual_budget(w), row=row)      # >= 0.9999 of a row (measured: every ray of every row)
        assert p > min_psnr, "row %d: PSNR %.2f dB" % (row, p)


# ---------------------------------------------------------------------------------------------
# BASELINE config 4: 800x800, threshold 0.1, image-tile shard over 8 ranks (balanced 5-row strips)
# ---------------------------------------------------------------------------------------------

def test_config4_thr01_eight_way_shard(classroom, tmp_path_factory):
    """Runs in the opt-in guarded two-precision selection and checks the full-size rows against the oracle in it; the host's default
    (the split engine on every ray) must select the same samples on every ray."""
    z, meta, sc, wts = classroom
    sc4 = dataclasses.replace(sc, threshold=0.1)
    d = _dir(tmp_path_factory, sc4, wts, "config4")
    w = h = 800
    world = 8
    strip = sharding.balanced_strip_rows(h, world)
    assert strip == 5
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16", sampling="guarded") as r:
        r.set_camera(z["pose"], z["rot"])
        assert abs(r.info.threshold - 0.1) < 1e-7
        rgb, full, st = r.render_numpy()
        cnt = r.buffer(R.BUF_RAY_COUNTS, np.int32, (w * h,))
    assert cnt.min() >= 1 and cnt.max() <= 8 and st.total_samples == int(cnt.sum())
    assert r._opt.sampling_mode == R.SAMPLING_MODES["guarded"] and st.rays_refined > 0 and st.guard_violations == 0 and st.guard_audit_mismatch == 0
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16") as rs:      # the default: split precision on every ray
        assert rs._opt.sampling_mode == R.SAMPLING_MODES["split"]
        rs.set_camera(z["pose"], z["rot"])
        rs.render(None, None, stats=True)
        assert np.array_equal(rs.buffer(R.BUF_RAY_COUNTS, np.int32, (w * h,)), cnt)
    img, samples = render_sharded(d, w, h, z["pose"], z["rot"], world, strip, precision="bf16", sampling="guarded")
    assert np.array_equal(img, full)                       # assembled bytes == unsharded frame
    assert sum(samples) == st.total_samples
    record("config4_shard", samples_per_rank=samples, imbalance=max(samples) / (sum(samples) / world))
    assert max(samples) / (sum(samples) / world) < 1.02    # interleaved strips spread the content evenly
    check_rows_against_oracle(sc4, wts, w, h, z["pose"], z["rot"], (250, 601), cnt, rgb, 0.999, 55.0, "config4_rows")


# ---------------------------------------------------------------------------------------------
# BASELINE config 3: 800x800 dense, 128 samples per ray (no compaction; composite_wave_kernel)
# ---------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def dense(tmp_path_factory):
    z, meta, sc = load_case("classroom_dense128")
    wts = case_weights(meta)
    return z, meta, sc, wts, _dir(tmp_path_factory, sc, wts, "dense")


@pytest.mark.parametrize("prec,min_psnr", [("bf16", 56.0), ("fp16", 75.0)])      # measured 61.9 / 80.5 dB
def test_dense_small_frame_16bit(dense, prec, min_psnr):
    z, meta, sc, wts, d = dense
    w, h = 48, 40
    ref = small_frame(dense, w, h)
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision=prec) as r:
        r.set_camera(z["pose"], z["rot"])
        rgb, rgba, st = r.render_numpy()
    assert st.total_samples == w * h * 128
    p = O.psnr(rgb, ref["rgb"])
    record("dense_small_frame", prec=prec, psnr_db=p, max_abs=float(np.abs(rgb - ref["rgb"]).max()))
    assert p > min_psnr, "PSNR %.2f dB" % p
    assert np.array_equal(rgba[:, :3], O.to_rgba8(rgb)[:, :3])


def test_dense_full_size_properties(dense):
    z, meta, sc, wts, d = dense
    w = h = 800
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16") as r:
        r.set_camera(z["pose"], z["rot"])
        assert r.info.dense == 1
        rgb, rgba, st = r.render_numpy()
        cnt = r.buffer(R.BUF_RAY_COUNTS, np.int32, (w * h,))
        off = r.buffer(R.BUF_RAY_OFFSETS, np.int32, (w * h,))
        tot = r.buffer(R.BUF_TOTAL, np.int32, (1,))
    assert st.total_samples == w * h * 128 == int(tot[0]) and (cnt == 128).all()
    assert np.array_equal(off.astype(np.int64), np.arange(w * h, dtype=np.int64) * 128)
    assert np.isfinite(rgb).all() and (rgba[:, 3] == 255).all()
    assert np.array_equal(rgba[:, :3], O.to_rgba8(rgb)[:, :3])
    for row in (77, 640):
        ref = O.render_frame(sc, wts, w, h, z["pose"], z["rot"], rows=(row, row + 1))
        sl = slice(row * w, (row + 1) * w)
        p = O.psnr(rgb[sl], ref["rgb"])
        record("dense_full_rows", row=row, psnr_db=p, max_abs=float(np.abs(rgb[sl] - ref["rgb"]).max()))
        assert p > 55.0, "row %d: PSNR %.2f dB" % (row, p)       # measured 59.3 / 66.2 dB


# ---------------------------------------------------------------------------------------------
# BASELINE config 5: 1920x1080 LLFF-NDC, fp16 shading, the thresholds the sweep test does not cover + 8-way shard
# ---------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def ndc(tmp_path_factory):
    z, meta, sc = load_case("ndc_synthetic_n8")
    wts = case_weights(meta)
    return z, meta, sc, wts, _dir(tmp_path_factory, sc, wts, "ndc")


@pytest.mark.parametrize("thr", [0.1, 0.3])
def test_config5_thresholds_01_03(ndc, thr):
    z, meta, sc, wts, d = ndc
    w, h = 1920, 1080
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="fp16", threshold=thr, sampling="guarded") as r:
        r.set_camera(z["pose"], z["rot"])
        rgb, rgba, st = r.render_numpy()
        cnt = r.buffer(R.BUF_RAY_COUNTS, np.int32, (w * h,))
        off = r.buffer(R.BUF_RAY_OFFSETS, np.int32, (w * h,))
        assert r.last_stats.sampling_overflow == 0
        # the opt-in guarded selection at this frame size: band silent, audit clean
        assert r._opt.sampling_mode == R.SAMPLING_MODES["guarded"] and st.rays_refined > 0 and st.guard_violations == 0 and st.guard_audit_mismatch == 0
    assert cnt.min() >= 1 and cnt.max() <= 8 and st.total_samples == int(cnt.sum())
    assert np.array_equal(off[1:].astype(np.int64), np.cumsum(cnt.astype(np.int64))[:-1])
    assert np.isfinite(rgb).all() and np.array_equal(rgba[:, :3], O.to_rgba8(rgb)[:, :3])
    sct = dataclasses.replace(sc, threshold=thr)
    check_rows_against_oracle(sct, wts, w, h, z["pose"], z["rot"], (200, 803), cnt, rgb, 0.999, 75.0, "config5_thr%.1f" % thr)   # measured 82.2 - 88.9 dB


def test_config5_eight_way_shard_1080p(ndc):
    z, meta, sc, wts, d = ndc
    w, h = 1920, 1080
    world = 8
    strip = sharding.balanced_strip_rows(h, world)
    assert strip == 5 and (h // strip) % world == 0
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="fp16", threshold=0.1) as r:
        r.set_camera(z["pose"], z["rot"])
        _, full, st = r.render_numpy()
    img, samples = render_sharded(d, w, h, z["pose"], z["rot"], world, strip, precision="fp16", threshold=0.1)
    assert np.array_equal(img, full) and sum(samples) == st.total_samples
    record("config5_shard", samples_per_rank=samples, imbalance=max(samples) / (sum(samples) / world))
    assert max(samples) / (sum(samples) / world) < 1.02


# ---------------------------------------------------------------------------------------------
# compositing multipliers: accumulationMult = weights / unset, and alpha under a losses[0] that keeps the oracle values
# out of compositing (reference-generated fixtures)
# ---------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def mult_cases(tmp_path_factory):
    out = {}
    for name in MULT_CASES:
        z, meta, sc = load_case(name)
        wts = case_weights(meta)
        out[name] = (z, meta, sc, wts, _dir(tmp_path_factory, sc, wts, name))
    return out


@pytest.mark.parametrize("name", MULT_CASES)
def test_composite_multiplier_modes_match_reference(mult_cases, name):
    z, meta, sc, wts, d = mult_cases[name]
    count, off, key, sw, sray, sbin = golden_samples(z, sc)
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, meta["w"], meta["h"])) as r:
        n = count.shape[0]
        rgb = r.empty((n, 3), np.float32)
        rgba = r.empty((n, 4), np.uint8)
        r.composite(r.to_device(z["shade_out"]), r.to_device(sw), r.to_device(off), r.to_device(count), n, rgb, rgba)
        out, out8 = rgb.numpy(), rgba.numpy()
    np.testing.assert_allclose(out, z["rgb"], rtol=0, atol=2e-6)       # reference output; expf vs torch.sigmoid
    exp8 = O.to_rgba8(z["rgb"])
    assert (np.abs(out8.astype(np.int16) - exp8.astype(np.int16)) <= 1).all() and (out8[:, 3] == 255).all()


@pytest.mark.parametrize("name", MULT_CASES)
def test_frame_multiplier_modes_match_oracle(mult_cases, name):
    z, meta, sc, wts, d = mult_cases[name]
    w, h = 112, 80
    ref = small_frame(mult_cases[name], w, h)
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="fp32") as r:
        r.set_camera(z["pose"], z["rot"])
        rgb, rgba, st = r.render_numpy()
        cnt, same = same_bin_sets(r, ref, w * h, sc.num_samples)
    check_identical(same, "frame_multiplier_modes", residual_budget(w * h), case=name)
    np.testing.assert_allclose(rgb[same], ref["rgb"][same], rtol=0, atol=3e-4)
    # the three modes really differ on this frame (a test that cannot tell them apart would be vacuous)
    other = dataclasses.replace(sc, accumulation_mult="alpha", losses0="NeRFWeightMultiplicationLoss")
    ref_alpha = O.render_rays(O.generate_ray_directions(w, h, sc.fov), z["pose"], z["rot"], other, wts, w, h)
    assert np.abs(ref_alpha["rgb"] - ref["rgb"]).max() > 0.05


# ---------------------------------------------------------------------------------------------
# the RCCL branch of bench.py on this box's one GPU: world size 1 under torch.distributed.run, backend nccl
# ---------------------------------------------------------------------------------------------

def test_bench_rccl_branch_world_size_one(tmp_path):
    """Process-group init on nccl (= RCCL), dist.gather of the strip payload, assemble_strips: the code path the driver's
    N > 1 runs take, with nothing swapped out -- only the world size differs."""
    one, dist1 = str(tmp_path / "one.npy"), str(tmp_path / "dist1.npy")
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-speed-mode"]
    a = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--dump-image", one] + common, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert a.returncode == 0, a.stderr[-2000:]
    env = dict(os.environ, ADANERF_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29741", "bench.py", "--gpus", "1",
                        "--dump-image", dist1] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    lines = [ln for ln in b.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, b.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert rec["config"]["exchange"]["backend"] == "nccl" and rec["config"]["exchange"]["rccl_ranks"] == 1
    assert rec["config"]["exchange"]["gathers"] == 3 + 1           # one RCCL gather per frame, warm-up included
    assert np.array_equal(np.load(one), np.load(dist1))


# ---------------------------------------------------------------------------------------------
# secondary outputs of the compositing step (depth_map = sum w z, acc_map = sum w)
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", AUX_CASES + ["classroom_dense128"])
def test_aux_outputs_match_oracle(name, tmp_path_factory):
    """adanerf_set_aux_outputs: adaptive (LDS-staged compositing kernel), dense (wave-per-ray kernel), FromClassifiedDepth
    (classic kernel), under each multiplier mode the fixtures carry; the oracle's values are pinned to the reference's
    NeRFOutputDepth / NeRFWeightsOutput in tests/test_oracle_golden.py."""
    z, meta, sc = load_case(name)
    wts = case_weights(meta)
    d = _dir(tmp_path_factory, sc, wts, "aux_" + name)
    w, h = (40, 32) if sc.threshold == 0.0 and sc.sampler != "FromClassifiedDepth" else (112, 80)
    ref = O.render_rays(O.generate_ray_directions(w, h, sc.fov), z["pose"], z["rot"], sc, wts, w, h, keep=True)
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h, batch_size=3000), precision="fp32") as r:
        r.set_camera(z["pose"], z["rot"])
        depth, acc = r.empty((w * h,), np.float32), r.empty((w * h,), np.float32)
        r.set_aux_outputs(depth, acc)
        disp = r.empty((w * h,), np.float32)
        r.set_disp_output(disp)
        rgb, rgba, st = r.render_numpy()
        dm, am = depth.numpy(), acc.numpy()
        # disp_map (src/nerf_raymarch_common.py:61 / :138) is a function of the two maps: exactly that function of what was written
        with np.errstate(divide="ignore", invalid="ignore"):
            exp_disp = np.float32(1.0) / np.maximum(np.float32(1e-10), dm / am)
        assert np.array_equal(disp.numpy(), exp_disp, equal_nan=True)
        r.set_aux_outputs(None, None)                                             # disparity alone: the maps come from scratch buffers
        disp.upload(np.zeros(w * h, np.float32))
        r.render_numpy()
        assert np.array_equal(disp.numpy(), exp_disp, equal_nan=True)
        r.set_disp_output(None)
        depth.upload(np.full(w * h, -7.0, np.float32))
        disp.upload(np.full(w * h, -3.0, np.float32))
        rgb2, _, _ = r.render_numpy()
        assert (depth.numpy() == -7.0).all() and (disp.numpy() == -3.0).all() and np.array_equal(rgb, rgb2)   # switched off: untouched, same image
    same = np.ones(w * h, bool)
    if "bins" in ref:                                       # adaptive: compare rays whose selection agrees (SURVEY "Hard parts")
        cnt_ok = np.isclose(np.abs(rgb - ref["rgb"]).max(axis=1), 0, atol=3e-4)
        same = cnt_ok
        check_identical(same, "aux_outputs_" + name, residual_budget(w * h))      # oracle evaluated on this box: at most max(1, 1e-4 n) rays
    if sc.sampler == "FromClassifiedDepth":
        # where a bin's probability mass is ~0 the inverse CDF is ill-conditioned and an occasional sample lands at the other
        # edge of an empty bin (see test_pdf_sampler_matches_reference): robust bounds, < 0.5 % of rays may deviate
        assert np.quantile(np.abs(am - ref["acc_map"]), 0.995) < 1e-4 and np.abs(am - ref["acc_map"]).max() < 2e-2
        assert np.quantile(np.abs(dm - ref["depth_map"]), 0.995) < 5e-4
    else:
        np.testing.assert_allclose(am[same], ref["acc_map"][same], rtol=0, atol=1e-4)
        np.testing.assert_allclose(dm[same], ref["depth_map"][same], rtol=2e-5, atol=2e-4)
    assert np.ptp(ref["acc_map"]) > 0.05


# ---------------------------------------------------------------------------------------------
# other topologies than 8 x 256 / skip 4 and the raySampleInput oracle input (run-time-shaped fp32 kernels)
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", TOPOLOGY_CASES)
def test_generic_topologies_match_the_reference(name, tmp_path_factory):
    from test_gpu_parity import crop_rows, run_rows
    z, meta, sc = load_case(name)
    wts = case_weights(meta)
    d = _dir(tmp_path_factory, sc, wts, "topo_" + name)
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, meta["w"], meta["h"]), precision="bf16") as r:
        r.set_camera(z["pose"], z["rot"])
        assert r.info.n_in0 == sc.n_in0 and r.info.precision == R.PREC_BF16      # generic shading nets run on the 16-bit engine too
        orc = run_rows(r, meta, lambda f, n, b: r.sample_mlp(f, n, b, None), 128)
        feat = run_rows(r, meta, lambda f, n, b: r.ray_features(f, n, b, None), sc.n_in0)
    n = z["oracle_in"].shape[0]
    np.testing.assert_allclose(feat[:n, :90], z["oracle_in"][:, :90], rtol=0, atol=2e-3)
    if sc.ray_sample_input:
        blk = np.abs(feat[:n, 90:] - z["oracle_in"][:, 90:]).reshape(n, sc.ray_sample_input, -1)
        assert blk[..., :9].max() < 5e-6 and blk.max() < 2e-3          # the 2^9 band amplifies 1-ulp differences of the points
    np.testing.assert_allclose(orc, z["oracle_out"], rtol=0, atol=3e-4)
    cnt, bins, _ = O.select_adaptive(orc, sc.num_samples, sc.threshold)
    same = (cnt == z["sel_count"]) & (bins == z["sel_bins"]).all(axis=1)
    check_identical(same, "generic_topology_selection", 0, case=name)      # against the reference's own selection (fixture)
    # the sampling net on the two run-time-shaped engines: split-precision pairs (default) vs exact fp32 MFMA -- same selection
    if not sc.ray_sample_input:
        with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, meta["w"], meta["h"]), precision="bf16", sampling="fp32") as r:
            r.set_camera(z["pose"], z["rot"])
            orc32 = run_rows(r, meta, lambda f, n, b: r.sample_mlp(f, n, b, None), 128)
        assert 0 < np.abs(orc32 - orc).max() < 3e-5            # two different engines, 22-bit vs 24-bit operands
        c32, b32, _ = O.select_adaptive(orc32, sc.num_samples, sc.threshold)
        assert (c32 == cnt).all() and (b32 == bins).all()
    # whole small frames against the oracle, both precisions requested
    w, h = 96, 64
    ref = O.render_rays(O.generate_ray_directions(w, h, sc.fov), z["pose"], z["rot"], sc, wts, w, h, keep=True)
    # the sampling net of these models runs on the split-precision (raySampleInput: fp32) run-time-shaped kernel; the shading net in the precision asked for
    for prec, min_psnr in (("fp32", 90.0), ("fp16", 72.0), ("bf16", 50.0)):
        with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h, batch_size=2500), precision=prec) as r:
            r.set_camera(z["pose"], z["rot"])
            rgb, rgba, st = r.render_numpy()
        # batched: the count buffer holds the last batch only -> identify the rays whose selection agrees through the fp32 image
        if prec == "fp32":
            same = np.abs(rgb - ref["rgb"]).max(axis=1) < 3e-4
            check_identical(same, "generic_topology_frame", residual_budget(w * h), case=name)
        p = O.psnr(rgb[same], ref["rgb"][same])
        record("generic_topology_frame", case=name, prec=prec, psnr_db=p, max_abs=float(np.abs(rgb[same] - ref["rgb"][same]).max()),
               samples=int(st.total_samples), ref_samples=int(ref["count"].sum()))
        assert p > min_psnr, (prec, p)
        assert abs(int(st.total_samples) - int(ref["count"].sum())) <= residual_budget(w * h) * sc.num_samples


# ---------------------------------------------------------------------------------------------
# vanilla NeRF with hierarchical sampling (inFeatures [RayMarchFromPoses, RayMarchFromCoarse])
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", COARSE_FINE_CASES)
def test_coarse_fine_stages_match_the_reference(name, tmp_path_factory):
    """Every stage of the mode against the step-by-step run of the reference's own feature sets and models
    (oracle/gen_golden.py gen_coarse_fine): camera rays, coarse network outputs at the uniform depths, the merged coarse +
    fine depths, the fine network's outputs, colour."""
    from test_gpu_parity import crop_rows
    z, meta, sc = load_case(name)
    wts = case_weights(meta)
    d = _dir(tmp_path_factory, sc, wts, "cf_" + name)
    nc, nf = sc.num_samples_coarse, sc.num_samples
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, meta["w"], meta["h"]), precision="fp32") as r:
        r.set_camera(z["pose"], z["rot"])
        assert r.info.sampler_mode == R.SAMPLER_COARSE_FINE and r.info.num_samples == nc + nf and r.info.num_samples_coarse == nc
        assert r.info.n_in0 == 6 + 6 * sum(sc.pos_enc[0]) and r.info.use_ndc == int(sc.use_ndc)
        rows = crop_rows(meta)
        nmax = max(n for _, n, _ in rows)
        rays = r.empty((nmax, 8), np.float32)
        off, cnt, tot = r.empty((nmax,), np.int32), r.empty((nmax,), np.int32), r.empty((1,), np.int32)
        keyc, rawc = r.empty((nmax * nc,), np.uint32), r.empty((nmax * nc, 4), np.float32)
        key, sz, raw = r.empty((nmax * (nc + nf),), np.uint32), r.empty((nmax * (nc + nf),), np.float32), r.empty((nmax * (nc + nf), 4), np.float32)
        rgb = r.empty((nmax, 3), np.float32)
        got = dict(rays=[], rawc=[], z=[], raw=[], rgb=[])
        for first, n, stride in rows:
            r.sample_uniform(first, n, rays, off, cnt, keyc, tot)
            assert int(tot.numpy()[0]) == n * nc and (cnt.numpy()[:n] == nc).all()
            r.shade_mlp_coarse(rays, keyc, tot, n * nc, rawc)
            r.sample_from_coarse(rawc, rays, n, off, cnt, key, sz, tot)
            assert int(tot.numpy()[0]) == n * (nc + nf) and (cnt.numpy()[:n] == nc + nf).all()
            r.shade_mlp_z(rays, key, sz, tot, n * (nc + nf), raw)
            r.composite_classic(raw, sz, rays, n, nc + nf, rgb, None)
            got["rays"].append(rays.numpy()[:n:stride].copy())
            got["rawc"].append(rawc.numpy()[:n * nc].reshape(n, nc, 4)[::stride].copy())
            got["z"].append(sz.numpy()[:n * (nc + nf)].reshape(n, nc + nf)[::stride].copy())
            got["raw"].append(raw.numpy()[:n * (nc + nf)].reshape(n, nc + nf, 4)[::stride].copy())
            got["rgb"].append(rgb.numpy()[:n:stride].copy())
        got = {k: np.concatenate(v) for k, v in got.items()}
    np.testing.assert_allclose(got["rays"][:, 0:3], z["p"], rtol=0, atol=1e-7)          # the camera position
    np.testing.assert_allclose(got["rays"][:, 4:7], z["nds"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(got["rawc"].reshape(-1, 4), z["coarse_out"], rtol=1e-3, atol=1.5e-3)
    zr = z["z_world"]
    assert (np.diff(got["z"], axis=1) >= 0).all()                                         # merged in ascending order
    rel = np.abs(got["z"] - zr) / zr
    assert np.median(rel) < 2e-6 and (rel < 1e-4).mean() > 0.995                          # inverse CDF: ill-conditioned in empty bins
    ok = (rel < 1e-5).all(axis=1)
    record("coarse_fine_stages", rays_with_all_depths_equal=float(ok.mean()), median_rel_depth_err=float(np.median(rel)),
           rgb_max_err=float(np.abs(got["rgb"][ok] - z["rgb"][ok]).max()))
    assert ok.mean() > 0.97                       # measured 0.9766 (classroom) / 1.0 (NDC): rays with a fine sample in an empty bin (above)
    np.testing.assert_allclose(got["raw"][ok].reshape(-1, 4), z["shade_out"].reshape(-1, nc + nf, 4)[ok].reshape(-1, 4), rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(got["rgb"][ok], z["rgb"][ok], rtol=0, atol=3e-4)


@pytest.mark.parametrize("prec,min_psnr", [("fp32", 90.0), ("fp16", 58.0), ("bf16", 45.0)])      # measured 112 / 65.7 / 51.5 dB
def test_coarse_fine_frames_match_the_oracle(prec, min_psnr, tmp_path_factory):
    """Whole small frames of the mode (batched and not) against the oracle, incl. depth / accumulation maps; Nc = 24, Nf = 40."""
    import dataclasses
    z, meta, sc = load_case(COARSE_FINE_CASES[0])
    sc = dataclasses.replace(sc, num_samples_coarse=24, num_samples=40)
    wts = case_weights(meta)
    d = _dir(tmp_path_factory, sc, wts, "cf_frame")
    w, h = 72, 48
    ref = O.render_rays(O.generate_ray_directions(w, h, sc.fov), z["pose"], z["rot"], sc, wts, w, h)
    outs = []
    for bs in (-1, 1000):
        with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h, batch_size=bs), precision=prec) as r:
            r.set_camera(z["pose"], z["rot"])
            depth, acc = r.empty((w * h,), np.float32), r.empty((w * h,), np.float32)
            r.set_aux_outputs(depth, acc)
            rgb, rgba, st = r.render_numpy()
            assert st.total_samples == w * h * 64
            outs.append((rgb, rgba, depth.numpy().copy(), acc.numpy().copy()))
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))                    # batching does not change a bit
    rgb, rgba, depth, acc = outs[0]
    # classic compositing: a sample's alpha is a steep function of its density where the density is near 0 (last sample: a step),
    # so the low-precision bound is on the bulk of the rays (see test_pdf_frame_matches_oracle)
    err = np.abs(rgb - ref["rgb"]).max(axis=1)
    record("coarse_fine_frame", prec=prec, psnr_db=O.psnr(rgb, ref["rgb"]), q97=float(np.quantile(err, 0.97)), max_abs=float(err.max()))
    if prec == "fp32":
        assert err.max() < 5e-4 and np.abs(acc - ref["acc_map"]).max() < 5e-4
        assert np.quantile(np.abs(depth - ref["depth_map"]) / np.maximum(ref["depth_map"], 1e-3), 0.99) < 1e-3
    assert O.psnr(rgb, ref["rgb"]) > min_psnr
    assert np.array_equal(rgba[:, :3], O.to_rgba8(rgb)[:, :3]) and (rgba[:, 3] == 255).all()


def test_context_lifecycle_returns_all_device_memory(tmp_path_factory):
    """Create / render / destroy contexts of every mode (adaptive, dense, DONeRF, coarse/fine, generic topology; all precisions,
    profiling on and off) many times: the device's free memory must come back to where it started (adanerf_destroy frees every
    buffer, packed network and event the context ever allocated)."""
    import ctypes as C
    import dataclasses
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    cases = []
    for name in ("classroom_n8_thr02", "classroom_dense128", "classroom_pdf_n8", "classroom_coarse_fine_16_24", "syn_6x128_skip2"):
        z, meta, sc = load_case(name)
        wts = case_weights(meta)
        cases.append((z, _dir(tmp_path_factory, sc, wts, "life_" + name)))

    def cycle():
        for z, d in cases:
            for prec in ("bf16", "fp16", "fp32"):
                with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, 160, 120, batch_size=7000), precision=prec) as r:
                    r.set_camera(z["pose"], z["rot"])
                    r.set_profiling(prec == "fp16")
                    r.render_numpy()
                    r.render_numpy()

    cycle()                                   # first use: HIP's own caches (code objects, pools) fill up
    base = free_bytes()
    for _ in range(6):
        cycle()
    assert abs(free_bytes() - base) <= 8 << 20, (base, free_bytes())      # 90 contexts later: within 8 MiB


# ---------------------------------------------------------------------------------------------
# SURVEY 8f N4 residuals: multiDepthFeatures != 128 (the adaptive sampler over 64 / 100 depth cells)
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", BINS_CASES)
def test_fewer_depth_cells_match_the_reference(name, tmp_path_factory):
    """A sampling network with D < 128 outputs on the device (rows padded with absent bins, pack.cpp): raw outputs against the
    reference-generated fixture in every sampling mode, the absent bins marked, the selection (counts, bins, kept values) bit for bit
    through the frame path incl. the guarded two-precision mode, sample depths with cell_size = 1 / D, fp32 colours."""
    from test_gpu_parity import run_rows
    z, meta, sc = load_case(name)
    wts = case_weights(meta)
    D, N = sc.depth_bins, sc.num_samples
    d = _dir(tmp_path_factory, sc, wts, "bins_" + name)
    x0, y0, cw, ch, stride = meta["crop"]
    idx = ((y0 + stride * np.arange(ch))[:, None] * meta["w"] + (x0 + stride * np.arange(cw))[None, :]).reshape(-1)
    for smp, tol in (("split", 2e-4), ("fp32", 2e-4), ("fp16", 2e-2)):
        with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, meta["w"], meta["h"]), precision="bf16", sampling=smp) as r:
            r.set_camera(z["pose"], z["rot"])
            orc = run_rows(r, meta, lambda f, n, b: r.sample_mlp(f, n, b, None), 128)
        np.testing.assert_allclose(orc[:, :D], z["oracle_out"], rtol=0, atol=tol)
        assert (orc[:, D:] <= -9e29).all()
    for prec, smp in (("fp32", "split"), ("bf16", "guarded"), ("bf16", "split")):
        with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, meta["w"], meta["h"]), precision=prec, sampling=smp) as r:
            r.set_camera(z["pose"], z["rot"])
            rgb, _, st = r.render_numpy()
            n = meta["w"] * meta["h"]
            cnt = r.buffer(R.BUF_RAY_COUNTS, np.int32, (n,))
            off = r.buffer(R.BUF_RAY_OFFSETS, np.int32, (n,))
            key = r.buffer(R.BUF_SAMPLE_KEY, np.uint32, (int(st.total_samples),))
            assert st.guard_violations == 0 and st.guard_audit_mismatch == 0
        check_identical(cnt[idx] == z["sel_count"], "bins_counts", 0, case=name, mode=smp)
        assert int((key & 127).max()) < D
        bins = np.full((len(idx), N), -1, np.int16)
        for k, i in enumerate(idx):
            bins[k, :cnt[i]] = (key[off[i]:off[i] + cnt[i]] & 127).astype(np.int16)
        check_identical((bins == z["sel_bins"]).all(axis=1), "bins_bins", 0, case=name, mode=smp)
        if prec == "fp32":
            err = float(np.abs(rgb[idx] - z["rgb"]).max())
            record("fewer_depth_cells_frame", case=name, max_abs_rgb_err=err)
            assert err < 3e-4, err


# ---------------------------------------------------------------------------------------------
# SURVEY 8f N4 residuals: every rayMarchNormalization of the reference, a custom centre, a config without the key
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", NORM_CASES)
def test_normalisations_match_the_reference(name, tmp_path_factory):
    """nerf_get_normalization_function (src/nerf_raymarch_common.py:195-244) on the device, against reference-generated fixtures: the
    encoded shading inputs of the fixture's samples (identity slots tight, top band loose: un-normalised positions reach |8| and the
    2^9 band multiplies a 1-ulp difference by 4 000), the fp32 shading network on them, and the fixture's colours from an fp32 frame."""
    from test_gpu_parity import golden_ray_records, golden_samples, run_rows
    z, meta, sc = load_case(name)
    wts = case_weights(meta)
    d = _dir(tmp_path_factory, sc, wts, "norm_" + name)
    count, off, key, sw, sray, sbin = golden_samples(z, sc)
    m = z["shade_in"].shape[0]
    S = min(len(key), 2570)
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, meta["w"], meta["h"]), precision="fp32") as r:
        r.set_camera(z["pose"], z["rot"])
        d_rays, d_key = r.to_device(golden_ray_records(z, meta, sc)), r.to_device(key[:S])
        out = r.empty((m, sc.n_in1), np.float32)
        r.shade_features(d_rays, d_key, m, out)
        f = out.numpy()
        raw_d = r.empty((S, 4), np.float32)
        r.shade_mlp(d_rays, d_key, r.to_device(np.array([S], np.int32)), S, raw_d)
        raw = raw_d.numpy()
    np.testing.assert_allclose(f[:, :3], z["shade_in"][:, :3], rtol=1e-6, atol=3e-6)        # the normalised position itself
    np.testing.assert_allclose(f[:, :9], z["shade_in"][:, :9], rtol=0, atol=2e-5)
    np.testing.assert_allclose(f, z["shade_in"], rtol=0, atol=5e-3)
    np.testing.assert_allclose(raw, z["shade_out"][:S], rtol=1e-3, atol=2e-2)             # raw outputs reach |30|; top-band conditioning as above
    # a whole fp32 frame of the fixture's crop: rays rendered row by row through the C ABI
    x0, y0, cw, ch, stride = meta["crop"]
    cols = x0 + stride * np.arange(cw)
    got = []
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, meta["w"], meta["h"]), precision="fp32") as r:
        r.set_camera(z["pose"], z["rot"])
        rgb_full, _, st = r.render_numpy()
        cnt = r.buffer(R.BUF_RAY_COUNTS, np.int32, (meta["w"] * meta["h"],))
    idx = ((y0 + stride * np.arange(ch))[:, None] * meta["w"] + cols[None, :]).reshape(-1)
    check_identical(cnt[idx] == z["sel_count"], "normalisation_counts", 0, case=name)
    err = float(np.abs(rgb_full[idx] - z["rgb"]).max())
    record("normalisation_frame", case=name, max_abs_rgb_err=err, shade_in_max_err=float(np.abs(f - z["shade_in"]).max()))
    assert err < 2e-3, err


# ---------------------------------------------------------------------------------------------
# SURVEY 8f N4, encodings: posEncArgs other than 10-4 / 2-2 (catch-all 16-band slot layout on the run-time-shaped kernels)
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ENCODING_CASES)
def test_other_encodings_match_the_reference(name, tmp_path_factory):
    """Any F_pos-F_dir up to 16 bands (src/util/feature_encoding.py:54-73).  Stage by stage through the C ABI: the encoded
    oracle inputs and the sampling network against the reference-generated fixture; the encoded shading inputs and the
    shading network against the oracle on the fixture's samples; then whole frames.  With 16 bands the top band multiplies
    its argument by 32768, so a position that differs from the reference's by 4e-6 (the ray table's own tolerance) changes
    that band's sine by 0.1: for the extreme case the networks are checked on the DEVICE's own ray records / features
    (what the kernels consumed) rather than on the reference's, and the frame comparison is left to the moderate case."""
    from test_gpu_parity import crop_rows, golden_ray_records, golden_samples, run_rows
    z, meta, sc = load_case(name)
    wts = case_weights(meta)
    d = _dir(tmp_path_factory, sc, wts, "enc_" + name.replace("-", "_"))
    (fp0, fd0), (fp1, fd1) = sc.pos_enc
    extreme = max(fp0, fp1) > 12
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, meta["w"], meta["h"]), precision="bf16") as r:
        r.set_camera(z["pose"], z["rot"])
        assert (r.info.n_in0, r.info.n_in1) == (sc.n_in0, 6 + 6 * (fp1 + fd1)) and r.info.precision == R.PREC_BF16
        orc = run_rows(r, meta, lambda f, n, b: r.sample_mlp(f, n, b, None), 128)
        feat = run_rows(r, meta, lambda f, n, b: r.ray_features(f, n, b, None), sc.n_in0)
        rays = run_rows(r, meta, lambda f, n, b: r.ray_features(f, n, None, b), 8)
        # shading stage on the fixture's selection, from the device's own ray records
        count, off, key, sw, sray, sbin = golden_samples(z, sc)
        m = min(len(key), 2048)
        d_rays, d_key = r.to_device(rays), r.to_device(key[:m])
        sfeat_d = r.empty((m, 6 + 6 * (fp1 + fd1)), np.float32)
        r.shade_features(d_rays, d_key, m, sfeat_d)
        raw_d, tot = r.empty((m, 4), np.float32), r.to_device(np.array([m], np.int32))
        r.shade_mlp(d_rays, d_key, tot, m, raw_d, precision=R.PREC_FP32)
        sfeat, raw = sfeat_d.numpy(), raw_d.numpy()
    # 1. the encodings: device PE of the device's rays == numpy PE of the same rays (every band; the accurate sin / cos)
    p_dev, nds_dev = rays[:, 0:3], rays[:, 4:7]
    exp_feat = O.oracle_features(nds_dev, p_dev, sc)
    np.testing.assert_allclose(feat, exp_feat, rtol=0, atol=1e-3 if extreme else 2e-5)
    n = z["oracle_in"].shape[0]
    if not extreme:
        np.testing.assert_allclose(feat[:n], z["oracle_in"], rtol=0, atol=2e-3)            # and the reference's own features
    # 2. the sampling network in the catch-all layout: on the device's features, and (moderate case) against the reference
    np.testing.assert_allclose(orc, O.sampling_mlp(feat, wts.net0), rtol=0, atol=3e-4)
    if not extreme:
        np.testing.assert_allclose(orc, z["oracle_out"], rtol=0, atol=3e-4)
        cnt, bins, _ = O.select_adaptive(orc, sc.num_samples, sc.threshold)
        check_identical((cnt == z["sel_count"]) & (bins == z["sel_bins"]).all(axis=1), "encoding_selection", 0, case=name)
    # 3. shading inputs and network
    exp_sfeat = O.shading_inputs(p_dev, nds_dev, sray[:m], O.to_world_depth(O.bin_t(sbin[:m].astype(np.int64)), sc), sc)
    np.testing.assert_allclose(sfeat, exp_sfeat, rtol=0, atol=2e-3 if extreme else 1e-3)      # 2^11 band x a few ulp of the normalised position (measured 3.7e-4)
    np.testing.assert_allclose(raw, O.shading_mlp(sfeat, wts.net1, n_pos=3 + 6 * fp1), rtol=1e-4, atol=1.2e-3)
    # 4. frames
    if not extreme:
        w, h = 96, 64
        ref = O.render_rays(O.generate_ray_directions(w, h, sc.fov), z["pose"], z["rot"], sc, wts, w, h, keep=True)
        for prec, min_psnr in (("fp32", 80.0), ("fp16", 70.0), ("bf16", 50.0)):
            with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h, batch_size=2500), precision=prec) as r:
                r.set_camera(z["pose"], z["rot"])
                rgb, rgba, st = r.render_numpy()
            if prec == "fp32":
                same = np.abs(rgb - ref["rgb"]).max(axis=1) < 1e-3
                check_identical(same, "encoding_frame", residual_budget(w * h), case=name)
            p = O.psnr(rgb[same], ref["rgb"][same])
            record("encoding_frame", case=name, prec=prec, psnr_db=p)
            assert p > min_psnr, (prec, p)
            assert abs(int(st.total_samples) - int(ref["count"].sum())) <= residual_budget(w * h) * sc.num_samples


# ---------------------------------------------------------------------------------------------
# Guarded two-precision selection at BASELINE size: config 2 over several poses, config 4, config 5
# ---------------------------------------------------------------------------------------------

def _selection_of(r, n_rays):
    st = r.render(None, None, stats=True)
    cnt = r.buffer(R.BUF_RAY_COUNTS, np.int32, (n_rays,))
    key = r.buffer(R.BUF_SAMPLE_KEY, np.uint32, (int(st.total_samples),))
    return st, cnt, key


@pytest.mark.parametrize("workload", ["config2_poses", "config4", "config5_thr02"])
def test_guarded_selection_at_full_size(classroom, ndc, workload, tmp_path_factory):
    """The bit-exact part of the contract for the mode bench.py measures: at 800 x 800 (config 2 at four poses inside the view cell,
    config 4's threshold 0.1) and 1920 x 1080 NDC (config 5 shape) the guarded mode's sample counts and keys equal the
    split-precision engine's on every ray, with the band calibrated by the library and the monitor silent."""
    if workload == "config5_thr02":
        z, meta, sc, wts, d = ndc
        w, h = 1920, 1080
        poses = [(z["pose"], z["rot"])]
    else:
        z, meta, sc, wts = classroom
        sc = dataclasses.replace(sc, threshold=0.1) if workload == "config4" else sc
        d = _dir(tmp_path_factory, sc, wts, "guard_" + workload)
        w = h = 800
        c, size = np.array(sc.view_cell_center, np.float32), np.array(sc.view_cell_size, np.float32)
        poses = [(z["pose"], z["rot"])]
        if workload == "config2_poses":
            poses += [((c + 0.3 * size * np.array(o, np.float32)).astype(np.float32), O.camera_rotation(yaw, pitch))
                      for o, yaw, pitch in (((0.4, -0.3, 0.2), 10.0, -15.0), ((-0.45, 0.4, -0.3), 200.0, 20.0), ((0.1, 0.45, 0.4), 300.0, 0.0))]
    with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16", sampling="split") as rs, \
            adanerf_amd.NeuralRenderer(adanerf_amd.Settings(d, w, h), precision="bf16", sampling="guarded") as rg:
        for i, (pose, rot) in enumerate(poses):
            rs.set_camera(pose, rot)
            rg.set_camera(pose, rot)
            st_s, cnt_s, key_s = _selection_of(rs, w * h)
            st_g, cnt_g, key_g = _selection_of(rg, w * h)
            rg.lib.adanerf_get_info(rg.handle, rg.info)
            record("guarded_full_size", workload=workload, pose=i, rays=w * h, refined=int(st_g.rays_refined), eps=float(rg.info.guard_eps),
                   eps_pair=float(rg.info.guard_eps_pair), monitor_max_seen=float(st_g.guard_max_seen), monitor_pair_seen=float(st_g.guard_pair_seen),
                   violations=int(st_g.guard_violations), audited=int(st_g.guard_audited), audit_mismatch=int(st_g.guard_audit_mismatch),
                   samples=int(st_g.total_samples), band_source=R.GUARD_FROM[int(rg.info.guard_calib_source)])
            assert st_g.total_samples == st_s.total_samples and np.array_equal(cnt_g, cnt_s) and np.array_equal(key_g, key_s)
            assert st_g.guard_violations == 0 and st_g.guard_max_seen <= rg.info.guard_eps and st_g.guard_pair_seen <= rg.info.guard_eps_pair
            # cumulative; the hosts' default lets the audit fill pass 2's last round: at least a quarter of 1/16 of the decided rays per frame
            assert st_g.guard_audit_mismatch == 0 and st_g.guard_audited >= (i + 1) * 0.004 * w * h
            assert 0 < st_g.rays_refined < w * h
