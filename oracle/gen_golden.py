#!/usr/bin/env python3
"""Generates tests/golden/*.npz by IMPORTING the reference's PyTorch path from
/root/reference/src (build container only -- the reference never travels) and driving
``TrainConfig.inference`` (src/train_data.py:278-299) on seeded ray batches.

This script is ours; nothing of the reference is copied.  The six third-party modules the
reference imports at module top but never executes on the inference path are stubbed with
empty modules.  Also records the reference's own CPU timing for
BASELINE configs 1-3 into tests/golden/reference_cpu_timing.json (``--timing``).

Usage:  python oracle/gen_golden.py [--timing]
"""
import dataclasses
import argparse
import json
import math
import os
import sys
import time
import types
from types import SimpleNamespace

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)

import adanerf_oracle as O  # noqa: E402  (for the ONNX reader, scene parser and ray table only)


def import_reference():
    for m in ["configargparse", "cv2", "imageio", "transforms3d", "ptflops", "pyrtools",
              "onnx", "onnxruntime", "tqdm", "matplotlib", "matplotlib.pyplot"]:
        if m not in sys.modules:
            try:
                __import__(m)
            except Exception:
                mod = types.ModuleType(m)
                if m == "tqdm":
                    mod.tqdm = lambda x, *a, **k: x
                sys.modules[m] = mod
    sys.path.insert(0, os.path.join(REF, "src"))
    import torch  # noqa
    import features, models, nerf_raymarch_common, train_data  # noqa
    import util.depth_transformations as dt  # noqa
    return SimpleNamespace(torch=torch, features=features, models=models,
                           nrc=nerf_raymarch_common, train_data=train_data, dt=dt)


def make_config(scene: O.Scene, weights=None):
    n = scene.num_samples
    layers, widths, skips = [8, 8], [256, 256], ["", "auto"]
    if weights is not None:      # topology of the two networks as the training config states it (src/models.py:362-370)
        d0 = len([k for k in weights.net0 if k.endswith(".weight")])
        d1, sk = O.shading_topology(weights.net1, 3 + 6 * scene.pos_enc[1][0])
        layers = [d0, d1]
        widths = [int(weights.net0["layers.0.weight"].shape[0]), int(weights.net1["pts_linears.0.weight"].shape[0])]
        skips = ["", "auto" if (d1 == 8 and sk == [4]) else (str(sk[0]) if sk else "99")]      # (several skips: build_reference constructs the NeRF class itself)
    sampler = "FromClassifiedDepthAdaptiveNoDepthRange" if scene.use_ndc else "FromClassifiedDepthAdaptive"
    if scene.sampler == "FromClassifiedDepth":
        sampler = "FromClassifiedDepth"
    return SimpleNamespace(
        inFeatures=["SpherePosDir", "RayMarchFromPoses"], outFeatures=["Raw", "RGBARayMarch"],
        posEnc=["nerf", "nerf"],
        posEncArgs=["%d-%d" % scene.pos_enc[0], "%d-%d" % scene.pos_enc[1]],
        raySampleInput=[scene.ray_sample_input, 0], multiDepthFeatures=[scene.depth_bins, scene.depth_bins], multiDepthIgnoreValue=[1.01, 1.01],
        multiDepthWindowSize=[], activation=["relu", "nerf"], layers=layers, layerWidth=widths,
        skips=skips, losses=[scene.losses0, "MSE"],
        numRaymarchSamples=[n, n], rayMarchSampler=["none", sampler],
        rayMarchSamplingStep=[1.0 / scene.depth_bins, 1.0 / scene.depth_bins], rayMarchSamplingNoise=[0.0, 0.0],
        rayMarchNormalization=["InverseSqrtDistCentered", scene.normalization] if scene.normalization else [],
        rayMarchNormalizationCenter=list(scene.normalization_center), adaptiveSamplingThreshold=scene.threshold,
        accumulationMult=scene.accumulation_mult if scene.sampler != "FromClassifiedDepth" else None, zNear=[scene.z_near, scene.z_near],
        zFar=[scene.z_far, scene.z_far], trainWithGTDepth=False, deterministicSampling=False,
        useNDC=scene.use_ndc, perturb=False, device="cpu", storeFullData=True,
        depthTransform=scene.depth_transform)


class Wrapper:
    def __init__(self, d):
        self.d = d

    def get_batch_input(self, i):
        return self.d


def build_reference(R, scene: O.Scene, weights: O.Weights, w, h):
    torch = R.torch
    cfg = make_config(scene, weights)
    f_in, f_out = R.features.FeatureSet.get_sets(cfg, "cpu")
    view = SimpleNamespace(fov=scene.fov, focal=O.focal_from_fov(w, scene.fov),
                           view_cell_center=list(scene.view_cell_center),
                           view_cell_size=list(scene.view_cell_size))
    di = SimpleNamespace(w=w, h=h, view=view, depth_max=scene.max_depth,
                         depth_range=list(scene.depth_range), depth_range_warped=list(scene.depth_range),
                         depth_transform=R.dt.LogTransform if scene.depth_transform == "log" else R.dt.LinearTransform,
                         use_warped_depth_range=[True, True])
    for f in f_in:
        f.initialize(cfg, di, "cpu")
    tc = R.train_data.TrainConfig()
    tc.f_in, tc.f_out = f_in, f_out
    n_in = [f.n_feat for f in f_in]
    m0 = R.models.ModelSelection.getModel(cfg, n_in[0], scene.depth_bins, "cpu", 0)
    m1 = R.models.ModelSelection.getModel(cfg, n_in[1], 4, "cpu", 1)
    d1, sk = O.shading_topology(weights.net1, 3 + 6 * scene.pos_enc[1][0])
    if len(sk) > 1:
        # The NeRF class takes a LIST of skips (src/models.py:200, 226-228, 260-261); ModelSelection can only hand it one (it wraps
        # config.skips[i] into a one-element list, src/models.py:370).  A network with several is built from the class directly.
        m1 = R.models.NeRF(cfg.layers[1], cfg.layerWidth[1], n_in=n_in[1], n_out=4, skips=[str(x) for x in sk], use_viewdirs=True, net_idx=1, config=cfg)
    m0.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.net0.items()})
    m1.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.net1.items()})
    m0.eval()
    m1.eval()
    tc.models = [m0, m1]
    return tc


def run_reference(R, tc, dirs, pose, rot, chunk=8192):
    torch = R.torch
    K = R.features.FeatureSetKeyConstants
    acc = {}
    for s in range(0, dirs.shape[0], chunk):
        dc = dirs[s:s + chunk]
        batch = {"ImagePose": torch.from_numpy(pose[None].copy()),
                 "ImageRotation": torch.from_numpy(rot[None].copy()),
                 "RayDirectionsSamples": torch.from_numpy(dc[None].copy())}
        with torch.no_grad():
            outs, dicts = tc.inference(Wrapper(batch), gradient=False, is_inference=True)
        d0, d1 = dicts
        n = dc.shape[0]
        item = dict(feat0=d0[K.input_feature_batch], orc=d0[K.network_output],
                    p=d0[K.input_feature_ray_origins], nds=d0[K.input_feature_ray_directions],
                    rgb=outs[1], est_depth=d1[K.nerf_estimated_depth].reshape(-1), acc=d1[K.nerf_weights_output].sum(-1))
        raw = d1[K.network_output]
        zv = d1[K.nerf_input_feature_z_vals]
        f1 = d1[K.input_feature_batch]
        ow = d1[K.oracle_weights] if K.oracle_weights in d1 else torch.zeros((n, 1))
        if K.oracle_weights not in d1 and K.adaptive_sample_positions not in d1:   # FromClassifiedDepth: [R*N,4] raw, [R,N] z, no selection
            item["count"] = torch.full((n,), zv.shape[1], dtype=torch.int32)
            item["z"] = zv.reshape(-1)
            item["raw"] = raw
            item["feat1"] = f1
            item["wts_slot"] = ow
            item["z_slot"] = zv
        elif raw.dim() == 3:   # adaptive path restored to [R, N, 4] with zero fill / NaN markers
            fin = torch.isfinite(zv)
            item["count"] = fin.sum(1).to(torch.int32)
            item["z"] = zv[fin]
            item["raw"] = raw[fin]
            item["feat1"] = f1[fin]
            item["wts_slot"] = ow
            item["z_slot"] = torch.nan_to_num(zv, nan=0.0)
        else:                # dense path [R*128, ...]
            item["count"] = torch.full((n,), 128, dtype=torch.int32)
            item["z"] = zv.reshape(-1)
            item["raw"] = raw
            item["feat1"] = f1
            item["wts_slot"] = ow
            item["z_slot"] = zv
        for k, v in item.items():
            acc.setdefault(k, []).append(v.numpy())
    return {k: np.concatenate(v) for k, v in acc.items()}


def z_to_bins(z_slot, count, scene: O.Scene, n_max):
    """Recover the kept bin ids from the reference's z values (slot-major [R,N], active
    slots first in ascending depth): invert to_world and round to the bin centre."""
    z = z_slot.astype(np.float64)
    if scene.use_ndc:
        t = z
    elif scene.depth_transform == "log":
        d0, d1 = scene.depth_range
        t = np.log(np.maximum(z - d0 + 1.0, 1e-30)) / math.log(d1 - d0 + 1.0)
    else:
        d0, d1 = scene.depth_range
        t = (z - d0) / (d1 - d0)
    k = np.rint(t * float(scene.depth_bins) - 0.5).astype(np.int64)
    mask = np.arange(n_max)[None, :] < count[:, None]
    return np.where(mask, k, -1).astype(np.int16)


def classroom_scene(n, thr):
    return O.load_scene(os.path.join(REF, "adanerf_real_time_viewer", "sample_pavillon_16") + "/", n, thr)


def subset_dirs(w, h, fov, x0, y0, cw, ch, stride=1):
    """[ch, cw] rays starting at pixel (x0, y0), every ``stride``-th pixel in both axes."""
    dirs = O.generate_ray_directions(w, h, fov).reshape(h, w, 3)
    return np.ascontiguousarray(dirs[y0:y0 + ch * stride:stride, x0:x0 + cw * stride:stride].reshape(-1, 3))


ONLY = None


def save_case(name, scene, meta, dirs, pose, rot, ref, n_max, weights_tag):
    if ONLY is not None and name not in ONLY:
        return
    count = ref["count"].astype(np.int32)
    if scene.sampler == "FromClassifiedDepth":
        bins = np.zeros((count.shape[0], n_max), dtype=np.int16)
        wts = np.zeros((count.shape[0], n_max), dtype=np.float32)
    elif scene.threshold == 0.0:
        bins = np.repeat(np.arange(128, dtype=np.int16)[None], count.shape[0], 0)
        wts = ref["wts_slot"].astype(np.float32)
    else:
        bins = z_to_bins(ref["z_slot"], count, scene, n_max)
        wts = ref["wts_slot"].astype(np.float32)
        if wts.shape != bins.shape:
            # losses[0] != NeRFWeightMultiplicationLoss: the sampler's kept oracle values never reach the feature
            # dict (src/features.py:503); they are the oracle outputs at the kept bins
            # (after the sampler's transform; restated here, since the reference drops them -- informational, unused downstream)
            wts = np.where(bins >= 0, np.take_along_axis(O.oracle_transform(ref["orc"], scene.losses0), np.maximum(bins, 0).astype(np.int64), axis=1), 0).astype(np.float32)
    m = dict(meta)
    m.update(dict(view_cell_center=list(scene.view_cell_center), view_cell_size=list(scene.view_cell_size),
                  depth_range=list(scene.depth_range), fov=scene.fov, max_depth=scene.max_depth,
                  num_samples=scene.num_samples, threshold=scene.threshold, z_near=scene.z_near,
                  z_far=scene.z_far, use_ndc=scene.use_ndc, depth_transform=scene.depth_transform,
                  pos_enc=[list(scene.pos_enc[0]), list(scene.pos_enc[1])],
                  normalization=scene.normalization, accumulation_mult=scene.accumulation_mult,
                  **({"normalization_center": list(scene.normalization_center)} if scene.normalization_center else {}),
                  **({"depth_bins": scene.depth_bins} if scene.depth_bins != 128 else {}),
                  sampler=scene.sampler, losses0=scene.losses0, ray_sample_input=scene.ray_sample_input, weights=weights_tag))
    n_f = min(64, ref["feat0"].shape[0], max(4, 65536 // ref["feat0"].shape[1]))
    m_f = min(64, ref["feat1"].shape[0])
    raw = ref["raw"].astype(np.float32)
    if raw.shape[0] > 40000:      # dense: keep the first 32 rays' samples only
        raw = raw[:32 * 128]
    np.savez_compressed(
        os.path.join(GOLD, name + ".npz"),
        meta=np.frombuffer(json.dumps(m).encode(), dtype=np.uint8),
        pose=pose.astype(np.float32), rot=rot.astype(np.float32), ray_dirs=dirs.astype(np.float32),
        nds=ref["nds"].astype(np.float32), p=ref["p"].astype(np.float32),
        oracle_in=ref["feat0"][:n_f].astype(np.float32), oracle_out=ref["orc"].astype(np.float32),
        sel_count=count.astype(np.uint8), sel_bins=bins, sel_weight=wts,
        z_world=ref["z"][:16384].astype(np.float32), shade_in=ref["feat1"][:m_f].astype(np.float32),
        shade_out=raw, rgb=ref["rgb"].astype(np.float32),
        # secondary outputs of the compositing step: NeRFOutputDepth (from_world of the weight-averaged sample depth;
        # src/features.py:571-577) and the accumulated opacity sum(weights) (src/nerf_raymarch_common.py:137-139)
        est_depth=ref["est_depth"].astype(np.float32), acc=ref["acc"].astype(np.float32))
    sz = os.path.getsize(os.path.join(GOLD, name + ".npz"))
    print("wrote %s.npz (%d KB)  rays=%d samples=%d mean=%.2f" %
          (name, sz // 1024, count.shape[0], int(count.sum()), float(count.mean())))


def gen_coarse_fine(R, name="classroom_coarse_fine_16_24", ndc=False):
    """Vanilla NeRF with hierarchical sampling: inFeatures [RayMarchFromPoses, RayMarchFromCoarse].  The reference
    cannot run this through TrainConfig.inference -- RayMarchFromCoarse.postprocess (src/features.py:688) unpacks five of the six
    values nerf_raw2outputs returns -- so the fixture drives the same objects step by step, exactly as inference() would
    (src/train_data.py:278-299): f_in[0].batch -> model 0 -> f_in[0].postprocess -> f_in[1].batch(prev_outs) -> model 1, and
    then makes the nerf_raw2outputs call of that postprocess itself."""
    if ONLY is not None and name not in ONLY:
        return
    torch = R.torch
    K = R.features.FeatureSetKeyConstants
    nc, nf = (16, 24) if not ndc else (12, 20)
    if ndc:
        # forward-facing scene in normalised device coordinates (RayMarchFromPoses with useNDC, src/features.py:429-431): the
        # samplers place depths over [0, 1] of the NDC ray (linear transform over the depth range [0, 1]); other encodings too
        base = O.Scene(view_cell_center=(0.0, 0.0, 0.0), view_cell_size=(2.0, 2.0, 1.0), depth_range=(0.0, 1.0), fov=1.0, max_depth=1.0,
                       num_samples=nf, threshold=0.0, use_ndc=True, depth_transform="linear", pos_enc=((8, 3), (10, 4)),
                       normalization="None", z_near=0.02, z_far=0.98)
    else:
        base = classroom_scene(nf, 0.0)
    sc = dataclasses.replace(base, sampler="CoarseFine", num_samples_coarse=nc, losses0="MSE", accumulation_mult="")
    wts = O.synthetic_coarse_fine_weights(31 if not ndc else 37, pos_enc=sc.pos_enc, alpha_bias=1.5 if not ndc else -0.6)
    cfg = make_config(sc)
    cfg.inFeatures = ["RayMarchFromPoses", "RayMarchFromCoarse"]
    cfg.outFeatures = ["RGBARayMarch", "RGBARayMarch"]
    cfg.activation = ["nerf", "nerf"]
    cfg.skips = ["auto", "auto"]
    cfg.losses = ["MSE", "MSE"]
    cfg.numRaymarchSamples = [nc, nf]
    cfg.rayMarchSampler = ["LinearlySpacedZNearZFar", "none"]
    cfg.rayMarchNormalization = [sc.normalization, sc.normalization]
    cfg.accumulationMult = None
    f_in, f_out = R.features.FeatureSet.get_sets(cfg, "cpu")
    w, h = (400, 400) if not ndc else (480, 270)
    view = SimpleNamespace(fov=sc.fov, focal=O.focal_from_fov(w, sc.fov), view_cell_center=list(sc.view_cell_center),
                           view_cell_size=list(sc.view_cell_size))
    di = SimpleNamespace(w=w, h=h, view=view, depth_max=sc.max_depth, depth_range=list(sc.depth_range),
                         depth_range_warped=list(sc.depth_range),
                         depth_transform=R.dt.LogTransform if sc.depth_transform == "log" else R.dt.LinearTransform,
                         use_warped_depth_range=[False, False])
    for f in f_in:
        f.initialize(cfg, di, "cpu")
    models = []
    for i, net in enumerate([wts.net0, wts.net1]):
        m = R.models.ModelSelection.getModel(cfg, f_in[i].n_feat, 4, "cpu", i)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in net.items()})
        m.eval()
        models.append(m)
    pose = np.array(sc.view_cell_center, dtype=np.float32) + np.array([0.05, -0.1, 0.02], np.float32)
    rot = O.camera_rotation(100.0, 5.0) if not ndc else np.eye(3, dtype=np.float32)      # LLFF-style camera looking down -z
    crop = [120, 150, 24, 16, 7] if not ndc else [100, 60, 24, 16, 9]
    if ndc:
        full = O.generate_ray_directions(w, h, sc.fov).reshape(h, w, 3)
        dirs = np.ascontiguousarray(full[crop[1]:crop[1] + crop[3] * crop[4]:crop[4], crop[0]:crop[0] + crop[2] * crop[4]:crop[4]].reshape(-1, 3))
    else:
        dirs = subset_dirs(w, h, sc.fov, 120, 150, 24, 16, 7)
    batch = {"ImagePose": torch.from_numpy(pose[None].copy()), "ImageRotation": torch.from_numpy(rot[None].copy()),
             "RayDirectionsSamples": torch.from_numpy(dirs[None].copy())}
    with torch.no_grad():
        d0 = f_in[0].batch(batch, prev_outs=[], is_inference=True)
        d0[K.network_output] = models[0](d0[K.input_feature_batch])
        f_in[0].postprocess(d0, batch)
        d1 = f_in[1].batch(batch, prev_outs=[d0], is_inference=True)
        d1[K.network_output] = models[1](d1[K.input_feature_batch])
        n = dirs.shape[0]
        zv = d1[K.nerf_input_feature_z_vals]
        rgb, disp, accm, w1, depth_map, alpha = R.nrc.nerf_raw2outputs(d1[K.network_output].reshape(n, zv.shape[1], -1), zv,
                                                                        d1[K.nerf_input_feature_ray_directions])
    meta = dict(w=w, h=h, crop=crop, yaw=100.0 if not ndc else 0.0, pitch=5.0 if not ndc else 0.0, view_cell_center=list(sc.view_cell_center),
                view_cell_size=list(sc.view_cell_size), depth_range=list(sc.depth_range), fov=sc.fov, max_depth=sc.max_depth,
                num_samples=nf, num_samples_coarse=nc, threshold=0.0, z_near=sc.z_near, z_far=sc.z_far, use_ndc=bool(ndc),
                depth_transform=sc.depth_transform, pos_enc=[list(sc.pos_enc[0]), list(sc.pos_enc[1])], normalization=sc.normalization,
                accumulation_mult="", sampler="CoarseFine", losses0="MSE", ray_sample_input=0,
                weights="synthetic_coarse_fine:31:1.5" if not ndc else "synthetic_coarse_fine:37:-0.6")
    np.savez_compressed(
        os.path.join(GOLD, name + ".npz"), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
        pose=pose, rot=rot, ray_dirs=dirs.astype(np.float32),
        nds=d0[K.nerf_input_feature_ray_directions].numpy().astype(np.float32), p=d0[K.nerf_input_feature_ray_origins].numpy().astype(np.float32),
        z_coarse=d0[K.nerf_input_feature_z_vals].numpy().astype(np.float32), coarse_in=d0[K.input_feature_batch][:64].numpy().astype(np.float32),
        coarse_out=d0[K.network_output].numpy().astype(np.float32), coarse_weights=d0[K.nerf_weights_output].numpy().astype(np.float32),
        coarse_rgb=d0[K.postprocessed_network_output].numpy().astype(np.float32),
        z_world=zv.numpy().astype(np.float32), shade_in=d1[K.input_feature_batch][:64].numpy().astype(np.float32),
        shade_out=d1[K.network_output].numpy().astype(np.float32), rgb=rgb.numpy().astype(np.float32),
        depth_map=depth_map.numpy().astype(np.float32), acc=accm.numpy().astype(np.float32))
    print("wrote %s.npz  rays=%d  coarse %d + fine %d samples" % (name, n, nc, nf))


def gen_dataset_reader(R, name="dataset_reader"):
    """what the reference's DatasetInfo / FullyLoadedViewCellDataset (src/datasets.py:146-213, 263-287, 361-365,
    479-542) derive from a dataset directory.  A tiny directory is written here (the files travel inside the fixture as
    byte arrays), the reference's readers are run over it -- imageio is not installed, its imread is served by PIL, which
    is what imageio uses for PNG -- and every derived quantity adanerf_amd/evaluate.py needs is recorded."""
    if ONLY is not None and name not in ONLY:
        return
    import io
    import tempfile
    import zlib
    import struct
    from PIL import Image
    import datasets as ref_datasets
    rng = np.random.default_rng(11)
    w, h, n_frames = 12, 10, 3
    d = tempfile.mkdtemp(prefix="adanerf_ds_")
    info = {"view_cell_center": [0.783, -3.19, 1.39], "view_cell_size": [0.7, 0.7, 0.2], "camera_scale": 1.5,
            "camera_base_orientation": [[1, 0, 0], [0, 0, -1], [0, 1, 0]], "resolution": [w, h],
            "camera_angle_x": 1.1386263370513916, "flip_depth": True, "depth_distance_adjustment": False,
            "depth_ignore": 1e10, "depth_range": [0.15422, 8.35819], "depth_range_warped_log": [0.1, 0.9],
            "depth_range_warped_lin": [0.2, 0.8]}
    files = {"dataset_info.json": json.dumps(info, indent=1).encode()}
    frames = []
    os.makedirs(os.path.join(d, "test"))
    for i in range(n_frames):
        yaw = 0.7 * i + 0.2
        rot = np.array([[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]]) @ np.array(info["camera_base_orientation"], dtype=np.float64)
        m = np.eye(4)
        m[:3, :3] = rot
        m[:3, 3] = np.array(info["view_cell_center"]) + rng.uniform(-0.3, 0.3, 3)
        frames.append({"file_path": "./test/%05d" % i, "transform_matrix": m.tolist()})
        img = rng.integers(0, 256, (h, w, 4 if i != 1 else 3), dtype=np.uint8)      # frame 1 is RGB, the others RGBA
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format="PNG")
        files["test/%05d.png" % i] = buf.getvalue()
    files["transforms_test.json"] = json.dumps({"camera_angle_x": info["camera_angle_x"], "frames": frames}, indent=1).encode()
    for rel, data in files.items():
        with open(os.path.join(d, rel), "wb") as f:
            f.write(data)
    sys.modules["imageio"].imread = lambda fn: np.asarray(Image.open(fn))
    ref_datasets.imageio = sys.modules["imageio"]
    sc = classroom_scene(8, 0.2)
    cfg = make_config(sc)
    cfg.data, cfg.scale, cfg.useNerfDepthMap, cfg.samplePlacementDir = d, 1, False, None
    f_in, f_out = R.features.FeatureSet.get_sets(cfg, "cpu")
    tc = R.train_data.TrainConfig()
    tc.f_in, tc.f_out = f_in, f_out
    di = ref_datasets.DatasetInfo(cfg, tc)
    ds = ref_datasets.FullyLoadedViewCellDataset(cfg, tc, di, set_name="test")
    out = dict(w=np.int32(di.w), h=np.int32(di.h), fov=np.float64(di.view.fov), focal=np.float64(di.view.focal),
               view_cell_center=np.array(di.view.view_cell_center, np.float64), view_cell_size=np.array(di.view.view_cell_size, np.float64),
               camera_scale=np.float64(di.view.camera_scale), base_rotation=np.array(di.view.base_rotation, np.float64),
               depth_range=np.array(di.depth_range, np.float64), depth_range_warped=np.array(di.depth_range_warped, np.float64),
               depth_max=np.float64(di.depth_max), poses=ds.poses.numpy().astype(np.float32), rotations=ds.rotations.numpy().astype(np.float32),
               color_images=ds.color_images.numpy().astype(np.float32), directions=ds.directions.numpy().astype(np.float32),
               image_names=np.frombuffer("\n".join(os.path.relpath(f, d) for f in ds.image_filenames).encode(), dtype=np.uint8))
    for rel, data in files.items():
        out["file:" + rel] = np.frombuffer(data, dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("wrote %s.npz  %d frames %dx%d, focal %.4f" % (name, n_frames, di.w, di.h, di.view.focal))


def gen_ray_table(R, name="ray_table_ref"):
    """A1 pinned to the reference's OWN pixel-ray table: src/util/raygeneration.py:10-26 called the way
    src/datasets.py:182, 268 calls it (focal = .5 w / tan(.5 fov); float64, cast to float32 by the dataset), for the frame sizes of the
    BASELINE configurations.  Whole rows are kept so that the fixture stays small: first, two middle and last row, plus the first and the
    last column."""
    if ONLY is not None and name not in ONLY:
        return
    from util.raygeneration import generate_ray_directions as ref_dirs
    out = {}
    for (w, h, fov) in ((800, 800, 1.1386263370513916), (400, 400, 1.1386263370513916), (1920, 1080, 1.0), (12, 10, 1.1386263370513916), (97, 61, 0.6)):
        focal = float(.5 * w / np.tan(.5 * fov))
        d = ref_dirs(w, h, fov, focal).astype(np.float32)          # [h, w, 3]
        rows = sorted({0, h // 2 - 1, h // 2, h - 1})
        key = "%dx%d" % (w, h)
        out[key + "/fov"] = np.float64(fov)
        out[key + "/rows"] = np.array(rows, np.int32)
        out[key + "/row_dirs"] = d[rows]
        out[key + "/col_dirs"] = d[:, [0, w - 1]]
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("wrote %s.npz  %s" % (name, sorted(k for k in out if k.endswith("/rows"))))


def gen_selection_edge_cases(R):
    """Synthetic oracle rows through the reference sampler alone
    (FromClassifiedDepthAdaptive.generate, src/nerf_raymarch_common.py:699-757):
    none >= thr, exact-threshold values, > N above thr, ties inside the top-N (not at the
    cutoff), plus random rows.  Rows whose tie straddles the N-cutoff are excluded (the
    reference's unstable sort makes them implementation-defined)."""
    torch = R.torch
    rng = np.random.default_rng(1234)
    out = {}
    for n_max, thr in [(1, 0.3), (4, 0.15), (8, 0.2), (16, 0.15), (32, 0.1)]:
        rows = []
        rows.append(rng.uniform(-0.5, thr - 0.01, size=(64, 128)))                    # none >= thr
        a = rng.uniform(-0.5, 1.8, size=(128, 128))                                    # random
        rows.append(a)
        b = rng.uniform(-0.5, thr - 0.05, size=(64, 128))                              # exact thr
        for i in range(64):
            idx = rng.choice(128, size=rng.integers(1, n_max + 1), replace=False)
            b[i, idx] = thr
        rows.append(b)
        rows.append(rng.uniform(thr + 0.01, 1.8, size=(64, 128)))                     # all above thr
        c = rng.uniform(-0.5, 0.1, size=(64, 128))                                    # ties inside top-N
        for i in range(64):
            if n_max >= 3:
                idx = rng.choice(128, size=3, replace=False)
                c[i, idx[:2]] = 1.5
                c[i, idx[2]] = 1.7
        rows.append(c)
        d = rng.uniform(-0.5, 1.8, size=(128, 128)) * (rng.uniform(size=(128, 128)) < 0.08)  # sparse peaks
        rows.append(d)
        orc = np.concatenate(rows).astype(np.float32)
        # drop rows with a tie straddling the N cutoff
        srt = -np.sort(-orc, axis=1)
        ok = np.ones(orc.shape[0], dtype=bool)
        if n_max < 128:
            ok &= ~((srt[:, n_max - 1] == srt[:, n_max]) & (srt[:, n_max - 1] >= np.float32(thr)))
        ok &= ~((srt[:, 0] == srt[:, 1]) & (srt[:, 0] < np.float32(thr)))            # arg-max tie in 'none' rows
        orc = orc[ok]
        scene = O.Scene((0, 0, 0), (1, 1, 1), (0.25, 9.5), 1.0, 10.0, n_max, thr)
        cfg = make_config(scene)
        smp = R.nrc.FromClassifiedDepthAdaptive(0.001, 1.0, n_max, 1 / 128.0, 0.0, config=cfg, net_idx=1)
        z, probs = smp.generate(orc.shape[0], "cpu", depth=torch.from_numpy(orc.copy()),
                                depth_range=list(scene.depth_range), depth_transform=R.dt.LogTransform)
        z = z.numpy()
        count = np.isfinite(z).sum(1).astype(np.int32)
        bins = z_to_bins(np.where(np.isfinite(z), z, 0.0), count, scene, n_max)
        key = "n%d" % n_max
        out[key + "_orc"] = orc
        out[key + "_thr"] = np.float32(thr)
        out[key + "_count"] = count.astype(np.uint8)
        out[key + "_bins"] = bins
        out[key + "_weight"] = probs.numpy().astype(np.float32)
        out[key + "_z"] = np.where(np.isfinite(z), z, 0.0).astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "selection_edge_cases.npz"), **out)
    print("wrote selection_edge_cases.npz (%d KB)" % (os.path.getsize(os.path.join(GOLD, "selection_edge_cases.npz")) // 1024))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--timing", action="store_true")
    ap.add_argument("--only", default=None, help="comma-separated case names: write only these fixtures")
    ap.add_argument("--out", default=None, help="write into this directory instead of tests/golden (tests/test_golden_regen.py)")
    args = ap.parse_args()
    global ONLY, GOLD
    ONLY = set(args.only.split(",")) if args.only else None
    if args.out:
        GOLD = os.path.abspath(args.out)
    os.makedirs(GOLD, exist_ok=True)
    R = import_reference()
    torch = R.torch
    torch.manual_seed(0)

    classroom = os.path.join(REF, "adanerf_real_time_viewer", "sample_pavillon_16") + "/"
    barber = os.path.join(REF, "adanerf_real_time_viewer", "sample") + "/"
    w_class = O.load_weights(classroom)
    w_barb = O.load_weights(barber)

    # real exported weights travel as data fixtures (the GPU box has no /root/reference)
    for tag, wts in [("sample_pavillon_16", w_class), ("sample", w_barb)] if not args.only else []:
        d = {"n0/" + k: v for k, v in wts.net0.items()}
        d.update({"n1/" + k: v for k, v in wts.net1.items()})
        np.savez_compressed(os.path.join(GOLD, "weights_%s.npz" % tag), **d)
        sc0 = O.load_scene(classroom if tag == "sample_pavillon_16" else barber)
        with open(os.path.join(GOLD, "scene_%s.json" % tag), "w") as f:
            json.dump(dict(view_cell_center=sc0.view_cell_center, view_cell_size=sc0.view_cell_size,
                           depth_range=sc0.depth_range, fov=sc0.fov, max_depth=sc0.max_depth,
                           num_samples=sc0.num_samples, threshold=sc0.threshold), f, indent=1)

    # --- case A: classroom weights, 800x800 frame, 48x32 crop, N=8 thr .2 (BASELINE config 2)
    sc = classroom_scene(8, 0.2)
    pose = np.array(sc.view_cell_center, dtype=np.float32)
    rot = O.camera_rotation(100.0, 0.0)
    dirs = subset_dirs(800, 800, sc.fov, 8, 10, 48, 32, 16)
    tc = build_reference(R, sc, w_class, 800, 800)
    ref = run_reference(R, tc, dirs, pose, rot)
    save_case("classroom_n8_thr02", sc, dict(w=800, h=800, crop=[8, 10, 48, 32, 16], yaw=100.0, pitch=0.0),
              dirs, pose, rot, ref, 8, "sample_pavillon_16")

    # --- case B: classroom weights N=16 thr .15 (shipped config), off-centre pose
    sc = classroom_scene(16, 0.15)
    pose_b = (np.array(sc.view_cell_center) + np.array([0.2, -0.1, 0.05])).astype(np.float32)
    rot_b = O.camera_rotation(-80.0, 10.0)
    dirs = subset_dirs(400, 400, sc.fov, 4, 6, 32, 24, 12)
    tc = build_reference(R, sc, w_class, 400, 400)
    ref = run_reference(R, tc, dirs, pose_b, rot_b)
    save_case("classroom_n16_thr015", sc, dict(w=400, h=400, crop=[4, 6, 32, 24, 12], yaw=-80.0, pitch=10.0),
              dirs, pose_b, rot_b, ref, 16, "sample_pavillon_16")

    # --- case C: dense 128 (BASELINE config 3), small crop
    sc = classroom_scene(128, 0.0)
    dirs = subset_dirs(800, 800, sc.fov, 400, 400, 16, 8)
    tc = build_reference(R, sc, w_class, 800, 800)
    ref = run_reference(R, tc, dirs, pose, rot)
    save_case("classroom_dense128", sc, dict(w=800, h=800, crop=[400, 400, 16, 8], yaw=100.0, pitch=0.0),
              dirs, pose, rot, ref, 128, "sample_pavillon_16")

    # --- case D: barbershop weights N=4 thr .15
    sc = O.load_scene(barber)
    pose_d = np.array(sc.view_cell_center, dtype=np.float32)
    rot_d = O.camera_rotation(-80.0, 0.0)
    dirs = subset_dirs(800, 800, sc.fov, 5, 7, 48, 32, 16)
    tc = build_reference(R, sc, w_barb, 800, 800)
    ref = run_reference(R, tc, dirs, pose_d, rot_d)
    save_case("barbershop_n4_thr015", sc, dict(w=800, h=800, crop=[5, 7, 48, 32, 16], yaw=-80.0, pitch=0.0),
              dirs, pose_d, rot_d, ref, 4, "sample")

    # --- case E: synthetic weights, fixed 8 samples/ray (BASELINE config 1 plumbing): every
    #     oracle output >= thr, so each ray keeps exactly its top-8.
    sc = classroom_scene(8, 0.05)
    w_syn = O.synthetic_weights(0, oracle_bias=3.0, oracle_scale=0.05)
    dirs = subset_dirs(400, 400, sc.fov, 180, 190, 32, 32)
    tc = build_reference(R, sc, w_syn, 400, 400)
    ref = run_reference(R, tc, dirs, pose, rot)
    save_case("synthetic_fixed8", sc, dict(w=400, h=400, crop=[180, 190, 32, 32], yaw=100.0, pitch=0.0,
                                           syn=dict(seed=0, oracle_bias=3.0, oracle_scale=0.05)),
              dirs, pose, rot, ref, 8, "synthetic")

    # --- case F: NDC / linear depth / 2-2 oracle encoding (BASELINE config 5), synthetic weights
    sc = O.Scene(view_cell_center=(0.0, 0.0, 0.0), view_cell_size=(2.0, 2.0, 1.0), depth_range=(0.9, 12.0),
                 fov=1.0, max_depth=12.0, num_samples=8, threshold=0.2, use_ndc=True,
                 depth_transform="linear", pos_enc=((2, 2), (10, 4)), normalization="None")
    w_ndc = O.synthetic_weights(7, n_in0=30, oracle_bias=-0.55, oracle_scale=0.5)
    pose_f = np.array([0.1, -0.05, 0.2], dtype=np.float32)
    rot_f = np.eye(3, dtype=np.float32)   # LLFF-style camera looking down -z
    full = O.generate_ray_directions(480, 270, sc.fov).reshape(270, 480, 3)
    dirs = np.ascontiguousarray(full[100:132, 200:248].reshape(-1, 3))
    tc = build_reference(R, sc, w_ndc, 480, 270)
    ref = run_reference(R, tc, dirs, pose_f, rot_f)
    save_case("ndc_synthetic_n8", sc, dict(w=480, h=270, crop=[200, 100, 48, 32], yaw=0.0, pitch=0.0,
                                           syn=dict(seed=7, n_in0=30, oracle_bias=-0.55, oracle_scale=0.5)),
              dirs, pose_f, rot_f, ref, 8, "synthetic")

    # --- case G: DONeRF-style inverse-CDF sampler + classic sigma/delta compositing
    import dataclasses
    sc = dataclasses.replace(classroom_scene(8, 0.2), sampler="FromClassifiedDepth", losses0="BCEWithLogitsLoss",
                             accumulation_mult="")
    dirs = subset_dirs(800, 800, sc.fov, 8, 10, 48, 32, 16)
    tc = build_reference(R, sc, w_class, 800, 800)
    ref = run_reference(R, tc, dirs, pose, rot)
    save_case("classroom_pdf_n8", sc, dict(w=800, h=800, crop=[8, 10, 48, 32, 16], yaw=100.0, pitch=0.0),
              dirs, pose, rot, ref, 8, "sample_pavillon_16")

    # --- cases H, I, J: the other compositing multipliers (src/nerf_raymarch_common.py:123-133): accumulationMult =
    #     weights, unset, and alpha under a losses[0] that keeps the oracle values out of compositing (src/features.py:503)
    for name, mult, l0 in [("classroom_n8_mult_weights", "weights", "NeRFWeightMultiplicationLoss"),
                           ("classroom_n8_mult_none", "", "NeRFWeightMultiplicationLoss"),
                           ("classroom_n8_loss_mse", "alpha", "MSE")]:
        sc = dataclasses.replace(classroom_scene(8, 0.2), accumulation_mult=mult, losses0=l0)
        dirs = subset_dirs(800, 800, sc.fov, 20, 30, 24, 16, 32)
        tc = build_reference(R, sc, w_class, 800, 800)
        ref = run_reference(R, tc, dirs, pose, rot)
        save_case(name, sc, dict(w=800, h=800, crop=[20, 30, 24, 16, 32], yaw=100.0, pitch=0.0), dirs, pose, rot, ref, 8,
                  "sample_pavillon_16")

    # --- cases Q, R, S: other topologies than 8 x 256 / skip 4 through the reference's own model
    #     classes (BaseNet / NeRF, src/models.py:18-82, 199-277), and the raySampleInput oracle input (src/features.py:876-888)
    base = classroom_scene(8, 0.65)      # random-init outputs spread over [-0.8, 0.9]: ~6 of 128 clear 0.65
    for name, syn, rsi in [("syn_6x128_skip2", dict(seed=21, layers=[6, 6], widths=[128, 128], skip1=2, oracle_bias=0.1, oracle_scale=0.3), 0),
                           ("syn_d2w128_d3w256_skip1", dict(seed=22, layers=[2, 3], widths=[128, 256], skip1=1, oracle_bias=0.1, oracle_scale=0.3), 0),
                           ("syn_rsi128_4x128", dict(seed=23, layers=[4, 8], widths=[128, 256], skip1=4, oracle_bias=0.1, oracle_scale=0.3), 128),
                           # widths the kernels have no instantiation for: run zero-padded to 128 / 256 and to 64 / 128 (pack.cpp pad_width);
                           # 70 // 2 = 35 rows in views_linears.0
                           ("syn_w96_w160_skip2", dict(seed=24, layers=[5, 6], widths=[96, 160], skip1=2, oracle_bias=0.1, oracle_scale=0.3), 0),
                           ("syn_w40_w70_skip1", dict(seed=25, layers=[3, 4], widths=[40, 70], skip1=1, oracle_bias=0.1, oracle_scale=0.3), 0),
                           # two skip connections in the NeRF trunk (layers 2 and 5 take cat([pts, h]))
                           ("syn_7x128_skips_1_4", dict(seed=26, layers=[4, 7], widths=[128, 128], skip1=[1, 4], oracle_bias=0.1, oracle_scale=0.3), 0)]:
        sc = dataclasses.replace(base, ray_sample_input=rsi)
        wts = O.synthetic_weights(syn["seed"], n_in0=sc.n_in0, oracle_bias=syn["oracle_bias"], oracle_scale=syn["oracle_scale"],
                                  layers=tuple(syn["layers"]), widths=tuple(syn["widths"]), skip1=syn["skip1"])
        dirs = subset_dirs(400, 400, sc.fov, 12, 20, 24, 16, 16)
        tc = build_reference(R, sc, wts, 400, 400)
        ref = run_reference(R, tc, dirs, pose, rot)
        save_case(name, sc, dict(w=400, h=400, crop=[12, 20, 24, 16, 16], yaw=100.0, pitch=0.0, syn=dict(syn, n_in0=sc.n_in0)),
                  dirs, pose, rot, ref, 8, "synthetic")

    # --- cases T, U (encodings): posEncArgs other than 10-4 / 2-2 (src/util/feature_encoding.py:54-73 accepts any
    #     band count; viewer config.cpp:142-146), default 8 x 256 topology: a mid-sized pair and the extremes the build supports
    for name, pe in [("syn_enc_6-3_12-2", ((6, 3), (12, 2))), ("syn_enc_16-1_1-16", ((16, 1), (1, 16)))]:
        sc = dataclasses.replace(classroom_scene(8, 0.65), pos_enc=pe)
        syn = dict(seed=31, oracle_bias=0.1, oracle_scale=0.3)
        wts = O.synthetic_weights(syn["seed"], n_in0=sc.n_in0, n_in1_pos=3 + 6 * pe[1][0], n_in1_dir=3 + 6 * pe[1][1],
                                  oracle_bias=syn["oracle_bias"], oracle_scale=syn["oracle_scale"])
        dirs = subset_dirs(400, 400, sc.fov, 12, 20, 24, 16, 16)
        tc = build_reference(R, sc, wts, 400, 400)
        ref = run_reference(R, tc, dirs, pose, rot)
        save_case(name, sc, dict(w=400, h=400, crop=[12, 20, 24, 16, 16], yaw=100.0, pitch=0.0, syn=dict(syn, n_in0=sc.n_in0)),
                  dirs, pose, rot, ref, 8, "synthetic")

    # --- cases V1..V8 (SURVEY 8f N4 residuals): every rayMarchNormalization the reference knows
    #     (nerf_get_normalization_function, src/nerf_raymarch_common.py:233-244) on the shading network's sample positions, a custom
    #     centre (rayMarchNormalizationCenter, src/features.py:460-467) and a config WITHOUT the key (-> MaxDepth, src/features.py:319-324);
    #     shipped classroom weights (trained for InverseSqrtDistCentered: other normalisations give it other inputs, which is all that matters here)
    for name, norm, centre in [("classroom_norm_none", "None", ()), ("classroom_norm_centered", "Centered", ()),
                               ("classroom_norm_maxdepth", "MaxDepth", ()), ("classroom_norm_maxdepthcentered", "MaxDepthCentered", ()),
                               ("classroom_norm_logcentered", "LogCentered", ()), ("classroom_norm_inversedistcentered", "InverseDistCentered", ()),
                               ("classroom_norm_isd_custom_centre", "InverseSqrtDistCentered", (0.5, -2.9, 1.2)),
                               ("classroom_norm_key_absent", "", ())]:
        sc = dataclasses.replace(classroom_scene(8, 0.2), normalization=norm, normalization_center=centre)
        dirs = subset_dirs(800, 800, sc.fov, 30, 44, 24, 16, 32)
        tc = build_reference(R, sc, w_class, 800, 800)
        ref = run_reference(R, tc, dirs, pose, rot)
        save_case(name, sc, dict(w=800, h=800, crop=[30, 44, 24, 16, 32], yaw=100.0, pitch=0.0), dirs, pose, rot, ref, 8, "sample_pavillon_16")

    # --- cases W1, W2 (SURVEY 8f N4 residuals): multiDepthFeatures != 128 -- a sampling network with 64 / 100 outputs, the adaptive
    #     sampler over that many depth cells (cell_size = 1 / D, src/nerf_raymarch_common.py:675-677, 726-741)
    for name, nb, n, thr in [("syn_bins64_n8", 64, 8, 0.65), ("syn_bins100_n6", 100, 6, 0.6)]:
        sc = dataclasses.replace(classroom_scene(n, thr), depth_bins=nb)
        syn = dict(seed=41 + nb, oracle_bias=0.1, oracle_scale=0.3, bins=nb)
        wts = O.synthetic_weights(syn["seed"], n_in0=sc.n_in0, oracle_bias=syn["oracle_bias"], oracle_scale=syn["oracle_scale"], bins=nb)
        dirs = subset_dirs(400, 400, sc.fov, 12, 20, 24, 16, 16)
        tc = build_reference(R, sc, wts, 400, 400)
        ref = run_reference(R, tc, dirs, pose, rot)
        save_case(name, sc, dict(w=400, h=400, crop=[12, 20, 24, 16, 16], yaw=100.0, pitch=0.0, syn=dict(syn, n_in0=sc.n_in0)),
                  dirs, pose, rot, ref, n, "synthetic")

    # --- case W3: the N4 residuals together -- 100 depth cells, hidden widths 96 / 160, two trunk skips, LogCentered positions
    sc = dataclasses.replace(classroom_scene(6, 0.6), depth_bins=100, normalization="LogCentered")
    syn = dict(seed=147, oracle_bias=0.1, oracle_scale=0.3, bins=100, layers=[4, 6], widths=[96, 160], skip1=[1, 3])
    wts = O.synthetic_weights(syn["seed"], n_in0=sc.n_in0, oracle_bias=syn["oracle_bias"], oracle_scale=syn["oracle_scale"], bins=100,
                              layers=tuple(syn["layers"]), widths=tuple(syn["widths"]), skip1=syn["skip1"])
    dirs = subset_dirs(400, 400, sc.fov, 12, 20, 24, 16, 16)
    tc = build_reference(R, sc, wts, 400, 400)
    ref = run_reference(R, tc, dirs, pose, rot)
    save_case("syn_combo_bins100_w96_w160_skips_logcentered", sc, dict(w=400, h=400, crop=[12, 20, 24, 16, 16], yaw=100.0, pitch=0.0, syn=dict(syn, n_in0=sc.n_in0)),
              dirs, pose, rot, ref, 6, "synthetic")

    # --- cases O, P: small crops that carry the secondary compositing outputs (all cases written from now on do)
    sc = classroom_scene(8, 0.2)
    dirs = subset_dirs(800, 800, sc.fov, 24, 40, 24, 16, 32)
    tc = build_reference(R, sc, w_class, 800, 800)
    ref = run_reference(R, tc, dirs, pose, rot)
    save_case("classroom_n8_aux", sc, dict(w=800, h=800, crop=[24, 40, 24, 16, 32], yaw=100.0, pitch=0.0), dirs, pose, rot, ref, 8,
              "sample_pavillon_16")
    sc = O.Scene(view_cell_center=(0.0, 0.0, 0.0), view_cell_size=(2.0, 2.0, 1.0), depth_range=(0.9, 12.0),
                 fov=1.0, max_depth=12.0, num_samples=8, threshold=0.2, use_ndc=True,
                 depth_transform="linear", pos_enc=((2, 2), (10, 4)), normalization="None")
    full = O.generate_ray_directions(480, 270, sc.fov).reshape(270, 480, 3)
    dirs = np.ascontiguousarray(full[60:76, 100:124].reshape(-1, 3))
    tc = build_reference(R, sc, w_ndc, 480, 270)
    ref = run_reference(R, tc, dirs, pose_f, rot_f)
    save_case("ndc_n8_aux", sc, dict(w=480, h=270, crop=[100, 60, 24, 16], yaw=0.0, pitch=0.0,
                                     syn=dict(seed=7, n_in0=30, oracle_bias=-0.55, oracle_scale=0.5)),
              dirs, pose_f, rot_f, ref, 8, "synthetic")

    # --- cases K, L: the oracle-output transforms of the other losses on the adaptive path
    #     (src/nerf_raymarch_common.py:686-690): sigmoid (BCEWithLogitsLoss) and softmax (CrossEntropyLoss) before the
    #     threshold test; the kept values then never reach compositing (src/features.py:503)
    for name, l0, thr in [("classroom_n8_bce_thr06", "BCEWithLogitsLoss", 0.6), ("classroom_n8_ce_thr0012", "CrossEntropyLoss", 0.012)]:
        sc = dataclasses.replace(classroom_scene(8, thr), losses0=l0)
        dirs = subset_dirs(800, 800, sc.fov, 20, 30, 32, 24, 24)
        tc = build_reference(R, sc, w_class, 800, 800)
        ref = run_reference(R, tc, dirs, pose, rot)
        save_case(name, sc, dict(w=800, h=800, crop=[20, 30, 32, 24, 24], yaw=100.0, pitch=0.0), dirs, pose, rot, ref, 8,
                  "sample_pavillon_16")

    # --- cases M, N: FromClassifiedDepth under NDC (linear depth through depth_range, NDC rays), and with the softmax transform
    sc = O.Scene(view_cell_center=(0.0, 0.0, 0.0), view_cell_size=(2.0, 2.0, 1.0), depth_range=(0.05, 0.95),
                 fov=1.0, max_depth=12.0, num_samples=8, threshold=0.2, use_ndc=True, depth_transform="linear",
                 pos_enc=((2, 2), (10, 4)), normalization="None", sampler="FromClassifiedDepth", losses0="BCEWithLogitsLoss",
                 accumulation_mult="")
    full = O.generate_ray_directions(480, 270, sc.fov).reshape(270, 480, 3)
    dirs = np.ascontiguousarray(full[100:124, 200:232].reshape(-1, 3))
    w_ndc_pdf = O.synthetic_weights(7, n_in0=30, oracle_bias=-0.55, oracle_scale=0.5, alpha_bias=2.3)   # some opaque, some empty rays
    tc = build_reference(R, sc, w_ndc_pdf, 480, 270)
    ref = run_reference(R, tc, dirs, pose_f, rot_f)
    save_case("ndc_pdf_n8", sc, dict(w=480, h=270, crop=[200, 100, 32, 24], yaw=0.0, pitch=0.0,
                                     syn=dict(seed=7, n_in0=30, oracle_bias=-0.55, oracle_scale=0.5, alpha_bias=2.3)),
              dirs, pose_f, rot_f, ref, 8, "synthetic")
    sc = dataclasses.replace(classroom_scene(8, 0.2), sampler="FromClassifiedDepth", losses0="CrossEntropyLoss", accumulation_mult="")
    dirs = subset_dirs(800, 800, sc.fov, 20, 30, 32, 24, 24)
    tc = build_reference(R, sc, w_class, 800, 800)
    ref = run_reference(R, tc, dirs, pose, rot)
    save_case("classroom_pdf_ce_n8", sc, dict(w=800, h=800, crop=[20, 30, 32, 24, 24], yaw=100.0, pitch=0.0), dirs, pose, rot, ref, 8,
              "sample_pavillon_16")

    if not args.only:
        gen_selection_edge_cases(R)
    gen_dataset_reader(R)
    gen_ray_table(R)
    gen_coarse_fine(R)
    gen_coarse_fine(R, "ndc_coarse_fine_12_20", ndc=True)

    if args.timing:
        timing = {"host": "build container", "threads": torch.get_num_threads(), "dtype": "fp32",
                  "note": "reference PyTorch path imported from /root/reference/src, chunk 8192, "
                          "median of repeats, shipped sample_pavillon_16 weights"}
        runs = []
        for (nm, n, thr, w, h, rows, reps) in [("config1_400x400_n8", 8, 0.2, 400, 400, 40, 3),
                                               ("config2_800x800_n8_thr0.2", 8, 0.2, 800, 800, 20, 3),
                                               ("config3_800x800_dense128", 128, 0.0, 800, 800, 1, 2)]:
            sc = classroom_scene(n, thr)
            tc = build_reference(R, sc, w_class, w, h)
            dirs = O.generate_ray_directions(w, h, sc.fov)[(h // 2) * w:(h // 2 + rows) * w]
            pose = np.array(sc.view_cell_center, dtype=np.float32)
            ts = []
            cnt = None
            for _ in range(reps):
                t0 = time.time()
                r = run_reference(R, tc, dirs, pose, O.camera_rotation(100.0, 0.0))
                ts.append(time.time() - t0)
                cnt = float(r["count"].mean())
            t = float(np.median(ts))
            runs.append(dict(config=nm, rays=int(dirs.shape[0]), seconds=t, rays_per_s=dirs.shape[0] / t,
                             mean_samples_per_ray=cnt,
                             extrapolated_s_per_frame=t * (w * h) / dirs.shape[0]))
            print(runs[-1])
        timing["runs"] = runs
        with open(os.path.join(GOLD, "reference_cpu_timing.json"), "w") as f:
            json.dump(timing, f, indent=1)


if __name__ == "__main__":
    main()
