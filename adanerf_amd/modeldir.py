"""Writers for the exported model-directory format (config.ini, dataset_info.txt, model{0,1}.onnx) and
small host helpers (camera rotation, seeded random-init weights).  Product-side tooling: bench.py and
the CLI examples use this; it shares no code with oracle/ (which has its own reader/writer for tests).

Format references: the reference's src/export.py:47-93 (dataset_info.txt + torch.onnx.export) and the viewer's
parser adanerf_real_time_viewer/src/config.cpp:200-344; minimal key set = sample_pavillon_16/config.ini.
"""
import math
import os
from typing import Dict

import numpy as np


def _vint(x: int) -> bytes:
    o = bytearray()
    while True:
        c = x & 0x7F
        x >>= 7
        if x:
            o.append(c | 0x80)
        else:
            o.append(c)
            return bytes(o)


def _ld(field: int, payload: bytes) -> bytes:
    return _vint((field << 3) | 2) + _vint(len(payload)) + payload


def write_onnx_initializers(path: str, tensors: Dict[str, np.ndarray]) -> None:
    """ONNX-shaped protobuf carrying graph.initializer entries only (ModelProto.graph = 7,
    GraphProto.initializer = 5, TensorProto dims = 1 / data_type = 2 (FLOAT) / name = 8 / raw_data = 9):
    what libadanerf_hip's loader reads from a real torch.onnx.export file."""
    graph = bytearray()
    for name, arr in tensors.items():
        a = np.ascontiguousarray(arr, dtype="<f4")
        t = bytearray()
        for d in a.shape:
            t += _vint((1 << 3) | 0) + _vint(int(d))
        t += _vint((2 << 3) | 0) + _vint(1)
        t += _ld(8, name.encode())
        t += _ld(9, a.tobytes())
        graph += _ld(5, bytes(t))
    with open(path, "wb") as f:
        f.write(_vint((1 << 3) | 0) + _vint(4) + _ld(7, bytes(graph)))


def write_model_dir(path: str, scene: dict, net0: Dict[str, np.ndarray], net1: Dict[str, np.ndarray]) -> None:
    """scene keys: view_cell_center[3], view_cell_size[3], depth_range[2], fov, max_depth, num_samples,
    threshold; optional use_ndc, depth_transform ('log'), pos_enc (((10,4),(10,4))), normalization,
    z_near, z_far, accumulation_mult; num_samples_coarse > 0 writes a vanilla-NeRF directory (inFeatures [RayMarchFromPoses,
    RayMarchFromCoarse], numRaymarchSamples [num_samples_coarse, num_samples]; net0 is then a NeRF net as well)."""
    os.makedirs(path, exist_ok=True)
    ndc = bool(scene.get("use_ndc", False))
    enc = scene.get("pos_enc", ((10, 4), (10, 4)))
    n = int(scene["num_samples"])
    sampler = "FromClassifiedDepthAdaptiveNoDepthRange" if ndc else "FromClassifiedDepthAdaptive"
    with open(os.path.join(path, "config.ini"), "w") as f:
        f.write("posEnc = [nerf, nerf]\n")
        f.write("posEncArgs = [%d-%d, %d-%d]\n" % (enc[0][0], enc[0][1], enc[1][0], enc[1][1]))
        nc = int(scene.get("num_samples_coarse", 0))
        if nc > 0:
            norm = scene.get("normalization", "InverseSqrtDistCentered")
            f.write("inFeatures = [RayMarchFromPoses, RayMarchFromCoarse]\noutFeatures = [RGBARayMarch, RGBARayMarch]\n")
            f.write("rayMarchSampler = [LinearlySpacedZNearZFar, none]\n")
            f.write("rayMarchNormalization = [%s, %s]\n" % (norm, norm))
            f.write("numRaymarchSamples = [%d, %d]\n" % (nc, n))
        else:
            f.write("inFeatures = [SpherePosDir, RayMarchFromPoses]\noutFeatures = [Raw, RGBARayMarch]\n")
            f.write("rayMarchSampler = [none, %s]\n" % sampler)
            f.write("rayMarchNormalization = [InverseSqrtDistCentered, %s]\n" % scene.get("normalization", "InverseSqrtDistCentered"))
            f.write("numRaymarchSamples = [%d, %d]\n" % (n, n))
        f.write("rayMarchSamplingStep = [0.0078125, 0.0078125]\nrayMarchSamplingNoise = [0.0, 0.0]\nraySampleInput = [0, 0]\n")
        f.write("depthTransform = %s\n" % scene.get("depth_transform", "log"))
        f.write("zNear = [%r, %r]\nzFar = [%r, %r]\n" % (scene.get("z_near", 0.001), scene.get("z_near", 0.001),
                                                       scene.get("z_far", 1.0), scene.get("z_far", 1.0)))
        f.write("adaptiveSamplingThreshold = %r\n" % float(scene["threshold"]))
        f.write("multiDepthFeatures = [128, 128]\nmultiDepthIgnoreValue = [1.01, 1.01]\n")
        f.write("accumulationMult = %s\n" % scene.get("accumulation_mult", "alpha"))
        f.write("useNDC = %s\n" % ("True" if ndc else "False"))
    with open(os.path.join(path, "dataset_info.txt"), "w") as f:
        f.write("view_cell_center = [%r, %r, %r]\n" % tuple(float(v) for v in scene["view_cell_center"]))
        f.write("view_cell_size = [%r, %r, %r]\n" % tuple(float(v) for v in scene["view_cell_size"]))
        f.write("depth_range = [%r, %r]\n" % tuple(float(v) for v in scene["depth_range"]))
        f.write("fov = %r\nfocal = 0.0\ncamera_scale = 1.0\nmax_depth = %r\n" % (float(scene["fov"]), float(scene["max_depth"])))
    write_onnx_initializers(os.path.join(path, "model0.onnx"), net0)
    write_onnx_initializers(os.path.join(path, "model1.onnx"), net1)


def camera_rotation(yaw_deg: float, pitch_deg: float) -> np.ndarray:
    """Row-major camera-to-world rotation for a z-up world, camera looking along -z with +y up; direction from
    yaw/pitch as the viewer's Camera::UpdateFeatureRot (adanerf_real_time_viewer/src/camera.cpp:143-158)."""
    y, p = math.radians(yaw_deg), math.radians(pitch_deg)
    fwd = np.array([math.cos(y) * math.cos(p), math.sin(y) * math.cos(p), math.sin(p)])
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    return np.stack([right, up, -fwd], axis=1).astype(np.float32)


def random_init_weights(seed: int = 0, n_in0: int = 90, n_pos: int = 63, n_dir: int = 27, oracle_bias: float = 0.1,
                        oracle_scale: float = 0.3, layers=(8, 8), widths=(256, 256), skip1: int = 4):
    """Seeded Kaiming-normal weights of the two architectures in the exported naming
    (layers.{0..D0-1}; pts_linears.{0..D1-1}, feature_linear, alpha_linear, views_linears.0, rgb_linear); ``layers`` / ``widths``
    / ``skip1``: depth and width of the two networks and the trunk's skip index (defaults = every shipped config)."""
    rng = np.random.default_rng(seed)

    def lin(n_out, n_in, scale=1.0):
        w = (rng.standard_normal((n_out, n_in)) * math.sqrt(2.0 / n_in) * scale).astype(np.float32)
        b = rng.uniform(-1.0 / math.sqrt(n_in), 1.0 / math.sqrt(n_in), size=(n_out,)).astype(np.float32)
        return w, b

    n0, n1 = {}, {}
    d0, w0, d1, w1 = layers[0], widths[0], layers[1], widths[1]
    dims = [n_in0] + [w0] * (d0 - 1) + [128]
    for i in range(d0):
        w, b = lin(dims[i + 1], dims[i], oracle_scale if i == d0 - 1 else 1.0)
        n0["layers.%d.weight" % i], n0["layers.%d.bias" % i] = w, (b + (oracle_bias if i == d0 - 1 else 0.0)).astype(np.float32)
    for i in range(d1):
        n1["pts_linears.%d.weight" % i], n1["pts_linears.%d.bias" % i] = lin(w1, n_pos if i == 0 else (w1 + n_pos if i == skip1 + 1 else w1))
    for nm, (o, k) in {"views_linears.0": (w1 // 2, w1 + n_dir), "feature_linear": (w1, w1), "alpha_linear": (1, w1),
                       "rgb_linear": (3, w1 // 2)}.items():
        n1[nm + ".weight"], n1[nm + ".bias"] = lin(o, k)
    return n0, n1


def random_init_nerf_pair(seed: int = 0, alpha_bias: float = 1.0):
    """Two seeded NeRF nets (8 x 256, skip 4, 10-4 encodings) for a vanilla-NeRF (coarse / fine) directory; the density
    bias keeps a random-init net from being transparent everywhere."""
    nets = []
    for i in range(2):
        _, n1 = random_init_weights(seed + 7919 * i)
        n1["alpha_linear.bias"] = (n1["alpha_linear.bias"] + np.float32(alpha_bias)).astype(np.float32)
        nets.append(n1)
    return nets[0], nets[1]
