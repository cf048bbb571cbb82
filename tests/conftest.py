import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_case(name):
    """Loads a golden fixture written by oracle/gen_golden.py; returns (npz dict, meta, Scene)."""
    import adanerf_oracle as O
    z = dict(np.load(os.path.join(GOLD, name + ".npz")))
    meta = json.loads(bytes(z.pop("meta")).decode())
    sc = O.Scene(view_cell_center=tuple(meta["view_cell_center"]), view_cell_size=tuple(meta["view_cell_size"]),
                 depth_range=tuple(meta["depth_range"]), fov=meta["fov"], max_depth=meta["max_depth"],
                 num_samples=meta["num_samples"], threshold=meta["threshold"], z_near=meta["z_near"],
                 z_far=meta["z_far"], use_ndc=meta["use_ndc"], depth_transform=meta["depth_transform"],
                 pos_enc=(tuple(meta["pos_enc"][0]), tuple(meta["pos_enc"][1])),
                 normalization=meta["normalization"], normalization_center=tuple(meta.get("normalization_center", ())),
                 accumulation_mult=meta["accumulation_mult"],
                 sampler=meta.get("sampler", "FromClassifiedDepthAdaptive"),
                 losses0=meta.get("losses0", "NeRFWeightMultiplicationLoss"), ray_sample_input=meta.get("ray_sample_input", 0),
                 num_samples_coarse=meta.get("num_samples_coarse", 0), depth_bins=meta.get("depth_bins", 128))
    return z, meta, sc


def case_weights(meta):
    """Weights for a golden case: shipped sample dirs are NOT available on the GPU box, so the
    real-weight cases carry their weights in tests/golden/weights_<tag>.npz (data fixtures
    extracted from the exported ONNX initializers by oracle/gen_golden.py)."""
    import adanerf_oracle as O
    tag = meta["weights"]
    if tag.startswith("synthetic_coarse_fine:"):       # two NeRF nets (vanilla coarse/fine mode): "synthetic_coarse_fine:<seed>:<alpha bias>"
        _, seed, ab = tag.split(":")
        return O.synthetic_coarse_fine_weights(int(seed), pos_enc=(tuple(meta["pos_enc"][0]), tuple(meta["pos_enc"][1])), alpha_bias=float(ab))
    if tag == "synthetic":
        s = meta["syn"]
        return O.synthetic_weights(s["seed"], n_in0=s.get("n_in0", 90), n_in1_pos=3 + 6 * meta["pos_enc"][1][0],
                                   n_in1_dir=3 + 6 * meta["pos_enc"][1][1], oracle_bias=s["oracle_bias"],
                                   oracle_scale=s["oracle_scale"], alpha_bias=s.get("alpha_bias", 0.0),
                                   layers=tuple(s.get("layers", (8, 8))), widths=tuple(s.get("widths", (256, 256))), skip1=s.get("skip1", 4),      # skip1: an index or a list of them
                                   bins=s.get("bins", 128))
    z = np.load(os.path.join(GOLD, "weights_%s.npz" % tag))
    n0 = {k[3:]: z[k] for k in z.files if k.startswith("n0/")}
    n1 = {k[3:]: z[k] for k in z.files if k.startswith("n1/")}
    return O.Weights(n0, n1)


def record(name, **values):
    """Appends one line of measured quantities (PSNR, agreement fractions, ...) to $ADANERF_MEASURED_LOG, if set: the
    tolerances stated in the tests are kept a margin below what this log shows (profiles/r02_parity_measured.log)."""
    path = os.environ.get("ADANERF_MEASURED_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(dict(test=name, **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in values.items()})) + "\n")


def check_identical(same, tag, max_residual=0, **ctx):
    """The bit-exact part of the contract (selected bins / sample counts per ray): `same` is the per-ray verdict.  Where the
    expected values are committed fixtures (fixed numbers) no ray may differ (max_residual = 0).  Where the oracle is
    evaluated on the test box's own CPU its sgemm's summation order is the host library's, and a ray whose N-th and
    (N+1)-th value differ by less than that noise may flip: callers allow max(1, 1e-4 n) such rays there.  The residual
    rays are recorded (ADANERF_MEASURED_LOG) and printed in the failure message."""
    same = np.asarray(same, dtype=bool)
    bad = np.flatnonzero(~same)
    record(tag + "_identical", rays=int(same.size), residual_rays=[int(i) for i in bad[:32]], n_residual=int(bad.size), **ctx)
    assert bad.size <= max_residual, "%s: %d of %d rays differ (allowed %d): rays %s" % (tag, bad.size, same.size, max_residual, bad[:32].tolist())


def residual_budget(n):
    """rays that may differ from an oracle computed on the test box's CPU: >= 0.9999 identical, at least one ray"""
    return max(1, int(1e-4 * n))


CASES = ["classroom_n8_thr02", "classroom_n16_thr015", "classroom_dense128", "barbershop_n4_thr015",
         "synthetic_fixed8", "ndc_synthetic_n8"]
# the compositing multipliers other than accumulationMult = alpha (src/nerf_raymarch_common.py:123-133, src/features.py:503)
MULT_CASES = ["classroom_n8_mult_weights", "classroom_n8_mult_none", "classroom_n8_loss_mse"]
# sigmoid / softmax applied to the sampling network's outputs before the adaptive selection (losses[0] = BCEWithLogitsLoss /
# CrossEntropyLoss, src/nerf_raymarch_common.py:686-690)
TRANSFORM_CASES = ["classroom_n8_bce_thr06", "classroom_n8_ce_thr0012"]
# FromClassifiedDepth beyond the DONeRF default: under NDC, and with the softmax transform
PDF_CASES = ["classroom_pdf_n8", "ndc_pdf_n8", "classroom_pdf_ce_n8"]
# fixtures that also carry the secondary compositing outputs (NeRFOutputDepth, accumulated opacity)
# topologies other than 8 x 256 / skip 4, and the raySampleInput oracle input
TOPOLOGY_CASES = ["syn_6x128_skip2", "syn_d2w128_d3w256_skip1", "syn_rsi128_4x128", "syn_w96_w160_skip2", "syn_w40_w70_skip1", "syn_7x128_skips_1_4"]
# positional encodings other than 10-4 / 2-2 (any posEncArgs up to 16 bands)
ENCODING_CASES = ["syn_enc_6-3_12-2", "syn_enc_16-1_1-16"]
# SURVEY 8f N4 residuals: every rayMarchNormalization the reference knows, a custom centre, and a config without the key
NORM_CASES = ["classroom_norm_none", "classroom_norm_centered", "classroom_norm_maxdepth", "classroom_norm_maxdepthcentered",
              "classroom_norm_logcentered", "classroom_norm_inversedistcentered", "classroom_norm_isd_custom_centre", "classroom_norm_key_absent"]
# multiDepthFeatures != 128: a sampling network with 64 / 100 depth cells (the device pads the rows to 128 absent bins)
BINS_CASES = ["syn_bins64_n8", "syn_bins100_n6", "syn_combo_bins100_w96_w160_skips_logcentered"]      # the last: with odd widths, two skips, LogCentered too
COARSE_FINE_CASES = ["classroom_coarse_fine_16_24", "ndc_coarse_fine_12_20"]      # vanilla NeRF, hierarchical sampling
AUX_CASES = ["classroom_n8_aux", "ndc_n8_aux", "classroom_n8_mult_weights", "classroom_n8_bce_thr06", "ndc_pdf_n8", "classroom_pdf_ce_n8"]
