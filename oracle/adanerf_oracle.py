"""CPU restatement of the AdaNeRF inference hot path (numpy, fp32).

TEST INFRASTRUCTURE ONLY.  Nothing under ``adanerf_amd/`` may import this
module: it is the checker for the HIP path (``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg are the only legal callers).

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference's own
PyTorch path (``/root/reference/src``) in the build container, drives
``TrainConfig.inference`` on seeded ray batches and commits the resulting
vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every
function below against them (indices exact, floats to the stated tolerance).

Each function cites the reference file:line it restates (paths relative to
``/root/reference``).  Arithmetic is float32 wherever the reference's is
(torch default dtype), float64 only where the reference uses numpy float64
(ray table).
"""
from __future__ import annotations

import dataclasses
import math
import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

F32 = np.float32
D_BINS = 128  # multiDepthFeatures (src/nerf_raymarch_common.py:710-712)


# --------------------------------------------------------------------------------------
# Exported model directory format (config.ini / dataset_info.txt / model{0,1}.onnx)
# --------------------------------------------------------------------------------------

def _parse_list(value: str) -> List[str]:
    """``[a, b]`` list syntax of adanerf_real_time_viewer/src/config.cpp:150-199."""
    s = value.strip()
    if s.startswith("["):
        s = s[1:s.index("]")]
    return [t.strip() for t in s.split(",")] if s.strip() != "" else []


def parse_kv_file(path: str) -> Dict[str, str]:
    """``key = value`` lines, whitespace stripped from both sides
    (adanerf_real_time_viewer/src/config.cpp:200-204, 284-295).  Section
    headers / comments have no '=' or are ignored by key lookup."""
    out: Dict[str, str] = {}
    with open(path, "r") as f:
        for line in f:
            if "=" not in line:
                continue
            k, v = line.split("=", 1)
            k = "".join(k.split())
            v = "".join(v.split())
            if k and not k.startswith(";") and not k.startswith("#"):
                out[k] = v
    return out


@dataclass
class Scene:
    """Scalar scene/config metadata: SURVEY §8a row A9
    (src/datasets.py:136-213, src/export.py:47-54, viewer include/config.h:10-63)."""
    view_cell_center: Tuple[float, float, float]
    view_cell_size: Tuple[float, float, float]
    depth_range: Tuple[float, float]        # warped range (dataset_info.txt `depth_range`)
    fov: float
    max_depth: float
    num_samples: int = 8                    # numRaymarchSamples[1]
    threshold: float = 0.2                  # adaptiveSamplingThreshold
    z_near: float = 0.001
    z_far: float = 1.0
    use_ndc: bool = False
    depth_transform: str = "log"            # "log" | "linear"
    pos_enc: Tuple[Tuple[int, int], Tuple[int, int]] = ((10, 4), (10, 4))
    normalization: str = "InverseSqrtDistCentered"   # rayMarchNormalization[1]; "" = the config has no such key (-> MaxDepth, src/features.py:319-324)
    normalization_center: Tuple[float, ...] = ()      # rayMarchNormalizationCenter: three values replace view_cell_center (src/features.py:460-467)
    accumulation_mult: str = "alpha"
    # "FromClassifiedDepthAdaptive" (AdaNeRF) or "FromClassifiedDepth" (DONeRF inverse-CDF sampling)
    sampler: str = "FromClassifiedDepthAdaptive"
    losses0: str = "NeRFWeightMultiplicationLoss"    # losses[0]: BCEWithLogitsLoss -> sigmoid on the oracle output
    ray_sample_input: int = 0                        # raySampleInput[0]: extra encoded points along the ray in the oracle input
    # sampler == "CoarseFine" (vanilla NeRF): inFeatures [RayMarchFromPoses, RayMarchFromCoarse],
    # rayMarchSampler [LinearlySpacedZNearZFar, none]; num_samples_coarse uniform samples for net 0 (a NeRF net as well),
    # num_samples more from its weights for net 1
    num_samples_coarse: int = 0
    depth_bins: int = 128                            # multiDepthFeatures: outputs of the sampling network = depth cells of the sampler (cell_size = 1 / D)

    @property
    def radius(self) -> float:
        # src/features.py:761  ||view_cell_size / 2||_2 (float64 numpy -> torch float64 tensor)
        return float(np.linalg.norm(np.array(self.view_cell_size, dtype=np.float64) / 2.0))

    @property
    def n_in0(self) -> int:
        fp, fd = self.pos_enc[0]      # src/features.py:738-740
        if self.sampler == "CoarseFine":
            return 3 + 6 * fp + 3 + 6 * fd        # src/features.py:622 (a RayMarch feature set in front of net 0 too)
        return (self.ray_sample_input * 3 + 3) * (2 * fp + 1) + 3 + 6 * fd

    @property
    def n_in1(self) -> int:
        fp, fd = self.pos_enc[1]
        return 3 + 6 * fp + 3 + 6 * fd


def load_scene(model_dir: str, num_samples: Optional[int] = None,
               threshold: Optional[float] = None) -> Scene:
    """config.ini then dataset_info.txt, later keys win
    (adanerf_real_time_viewer/src/config.cpp:270-344)."""
    kv = parse_kv_file(os.path.join(model_dir, "config.ini"))
    kv.update(parse_kv_file(os.path.join(model_dir, "dataset_info.txt")))
    fl = lambda k: [float(x) for x in _parse_list(kv[k])]
    enc = []
    for item in _parse_list(kv.get("posEncArgs", "[10-4,10-4]")):
        if item == "none":
            enc.append((4, 0))      # config.cpp:142-146
        else:
            a, b = item.split("-")
            enc.append((int(a), int(b)))
    norm = _parse_list(kv.get("rayMarchNormalization", "[]"))
    sc = Scene(
        view_cell_center=tuple(fl("view_cell_center")),
        view_cell_size=tuple(fl("view_cell_size")),
        depth_range=tuple(fl("depth_range")),
        fov=float(kv["fov"]),
        max_depth=float(kv["max_depth"]),
        num_samples=int(_parse_list(kv["numRaymarchSamples"])[-1]),
        threshold=float(kv.get("adaptiveSamplingThreshold", "-1")),
        z_near=float(_parse_list(kv.get("zNear", "[0.001,0.001]"))[-1]),
        z_far=float(_parse_list(kv.get("zFar", "[1.0,1.0]"))[-1]),
        use_ndc=kv.get("useNDC", "False") == "True",
        depth_transform=kv.get("depthTransform", "linear"),
        pos_enc=(enc[0], enc[1]),
        normalization=norm[-1] if norm else "",
        normalization_center=tuple(float(x) for x in _parse_list(kv.get("rayMarchNormalizationCenter", "[]"))),
        accumulation_mult=kv.get("accumulationMult", ""),
    )
    smp = _parse_list(kv.get("rayMarchSampler", "[none,FromClassifiedDepthAdaptive]"))
    sc.sampler = "FromClassifiedDepth" if smp[-1] == "FromClassifiedDepth" else "FromClassifiedDepthAdaptive"
    if _parse_list(kv.get("inFeatures", "[SpherePosDir,RayMarchFromPoses]")) == ["RayMarchFromPoses", "RayMarchFromCoarse"]:
        sc.sampler = "CoarseFine"      # vanilla NeRF: numRaymarchSamples = [Nc, Nf], zNear / zFar of net 0 place the coarse samples
        sc.num_samples_coarse = int(_parse_list(kv["numRaymarchSamples"])[0])
        sc.z_near = float(_parse_list(kv.get("zNear", "[0.001,0.001]"))[0])
        sc.z_far = float(_parse_list(kv.get("zFar", "[1.0,1.0]"))[0])
    ls = _parse_list(kv.get("losses", "[NeRFWeightMultiplicationLoss,MSE]"))
    sc.losses0 = ls[0] if ls else "NeRFWeightMultiplicationLoss"
    mdf = _parse_list(kv.get("multiDepthFeatures", "[128,128]"))
    sc.depth_bins = int(mdf[-1]) if mdf else 128
    rsi = _parse_list(kv.get("raySampleInput", "[0,0]"))
    sc.ray_sample_input = int(rsi[0]) if rsi else 0
    if num_samples is not None:
        sc.num_samples = num_samples
    if threshold is not None:
        sc.threshold = threshold
    return sc


# --- minimal protobuf reader for ONNX graph.initializer (src/export.py:78-83 writes these) ---

def _rv(b: bytes, i: int) -> Tuple[int, int]:
    r = 0
    s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def _fields(b: bytes):
    i = 0
    n = len(b)
    while i < n:
        k, i = _rv(b, i)
        f, w = k >> 3, k & 7
        if w == 0:
            v, i = _rv(b, i)
        elif w == 2:
            l, i = _rv(b, i)
            v = b[i:i + l]
            i += l
        elif w == 5:
            v = b[i:i + 4]
            i += 4
        elif w == 1:
            v = b[i:i + 8]
            i += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % w)
        yield f, w, v


def read_onnx_initializers(path: str) -> Dict[str, np.ndarray]:
    """ModelProto.graph(7) -> GraphProto.initializer(5) -> TensorProto{dims=1,
    data_type=2 (1=FLOAT), name=8, raw_data=9, float_data=4}.  PyTorch [out,in] layout."""
    with open(path, "rb") as f:
        b = f.read()
    out: Dict[str, np.ndarray] = {}
    for f1, w1, v1 in _fields(b):
        if f1 != 7 or w1 != 2:
            continue
        for f2, w2, v2 in _fields(v1):
            if f2 != 5 or w2 != 2:
                continue
            dims: List[int] = []
            name = None
            raw = None
            fdata: List[float] = []
            dt = 1
            for f3, w3, v3 in _fields(v2):
                if f3 == 1:
                    if w3 == 0:
                        dims.append(v3)
                    else:
                        j = 0
                        while j < len(v3):
                            d, j = _rv(v3, j)
                            dims.append(d)
                elif f3 == 2:
                    dt = v3
                elif f3 == 8:
                    name = v3.decode()
                elif f3 == 9:
                    raw = v3
                elif f3 == 4:
                    if w3 == 2:
                        fdata.extend(struct.unpack("<%df" % (len(v3) // 4), v3))
                    else:
                        fdata.append(struct.unpack("<f", v3)[0])
            if dt != 1 or name is None:
                continue
            if raw is not None:
                arr = np.frombuffer(raw, dtype="<f4").copy()
            else:
                arr = np.array(fdata, dtype=F32)
            out[name] = arr.reshape(dims if dims else ())
    return out


def write_onnx_initializers(path: str, tensors: Dict[str, np.ndarray]) -> None:
    """Writes a minimal ONNX-shaped protobuf that carries only graph.initializer
    entries (enough for every loader in this repo; used for synthetic-weight
    model directories in tests/bench).  Same field numbers as above."""
    def vint(x: int) -> bytes:
        o = bytearray()
        while True:
            c = x & 0x7F
            x >>= 7
            if x:
                o.append(c | 0x80)
            else:
                o.append(c)
                return bytes(o)

    def ld(fn: int, payload: bytes) -> bytes:
        return vint((fn << 3) | 2) + vint(len(payload)) + payload

    graph = bytearray()
    for name, arr in tensors.items():
        a = np.ascontiguousarray(arr, dtype="<f4")
        t = bytearray()
        for d in a.shape:
            t += vint((1 << 3) | 0) + vint(int(d))
        t += vint((2 << 3) | 0) + vint(1)
        t += ld(8, name.encode())
        t += ld(9, a.tobytes())
        graph += ld(5, bytes(t))
    model = vint((1 << 3) | 0) + vint(4) + ld(7, bytes(graph))
    with open(path, "wb") as f:
        f.write(model)


@dataclass
class Weights:
    """fp32 weights in PyTorch [out,in] layout, named as the exported ONNX names them."""
    net0: Dict[str, np.ndarray] = field(default_factory=dict)   # layers.{0..7}.{weight,bias}
    net1: Dict[str, np.ndarray] = field(default_factory=dict)   # pts_linears.*, alpha_linear, ...


def load_weights(model_dir: str) -> Weights:
    return Weights(read_onnx_initializers(os.path.join(model_dir, "model0.onnx")),
                   read_onnx_initializers(os.path.join(model_dir, "model1.onnx")))


def synthetic_weights(seed: int, n_in0: int = 90, n_in1_pos: int = 63, n_in1_dir: int = 27,
                      oracle_bias: float = 0.0, oracle_scale: float = 1.0, alpha_bias: float = 0.0,
                      layers: Tuple[int, int] = (8, 8), widths: Tuple[int, int] = (256, 256), skip1: int = 4, bins: int = D_BINS) -> Weights:
    """Seeded Kaiming-normal weights in the exported layout (nn.init.kaiming_normal_ as
    src/models.py:77-78, 246-250: std = sqrt(2/fan_in)); biases U(-1/sqrt(fan_in), ..) like
    nn.Linear's default.  ``oracle_bias`` is added to the sampling net's last bias so a
    chosen fraction of outputs clears the threshold (SURVEY §8d "W-syn").  ``layers`` / ``widths`` / ``skip1``: depth
    and width of the two networks and the NeRF trunk's skip index (src/models.py:18-82, 199-250; defaults = every
    shipped config)."""
    rng = np.random.default_rng(seed)

    def lin(n_out, n_in, scale=1.0):
        w = (rng.standard_normal((n_out, n_in)) * math.sqrt(2.0 / n_in) * scale).astype(F32, copy=False)
        bnd = 1.0 / math.sqrt(n_in)
        b = rng.uniform(-bnd, bnd, size=(n_out,)).astype(F32, copy=False)
        return w, b

    n0: Dict[str, np.ndarray] = {}
    d0, w0 = layers[0], widths[0]
    dims = [n_in0] + [w0] * (d0 - 1) + [bins]
    for i in range(d0):
        w, b = lin(dims[i + 1], dims[i], oracle_scale if i == d0 - 1 else 1.0)
        if i == d0 - 1:
            b = (b + oracle_bias).astype(F32, copy=False)
        n0["layers.%d.weight" % i] = w
        n0["layers.%d.bias" % i] = b
    n1: Dict[str, np.ndarray] = {}
    d1, w1 = layers[1], widths[1]
    for i in range(d1):
        sk = tuple(skip1) if isinstance(skip1, (tuple, list)) else (skip1,)      # the NeRF class takes a list of skips (src/models.py:200, 226-228)
        k = n_in1_pos if i == 0 else (w1 + n_in1_pos if (i - 1) in sk and i - 1 >= 0 else w1)
        w, b = lin(w1, k)
        n1["pts_linears.%d.weight" % i] = w
        n1["pts_linears.%d.bias" % i] = b
    for nm, (o, k) in {"views_linears.0": (w1 // 2, w1 + n_in1_dir), "feature_linear": (w1, w1),
                       "alpha_linear": (1, w1), "rgb_linear": (3, w1 // 2)}.items():
        w, b = lin(o, k)
        n1[nm + ".weight"] = w
        n1[nm + ".bias"] = b
    # classic sigma/delta compositing takes relu(density): shift it so that a random-init net is not transparent everywhere
    n1["alpha_linear.bias"] = (n1["alpha_linear.bias"] + F32(alpha_bias)).astype(F32, copy=False)
    return Weights(n0, n1)


def synthetic_coarse_fine_weights(seed: int, pos_enc=((10, 4), (10, 4)), alpha_bias: float = 0.0, layers: Tuple[int, int] = (8, 8),
                                 widths: Tuple[int, int] = (256, 256), skips: Tuple[int, int] = (4, 4)) -> Weights:
    """Two NeRF nets (src/models.py:199-277) for the coarse/fine mode: net 0 and net 1 both take [PE(pos) | PE(dir)]."""
    nets = []
    for i in range(2):
        fp, fd = pos_enc[i]
        w = synthetic_weights(seed + 7919 * i, n_in1_pos=3 + 6 * fp, n_in1_dir=3 + 6 * fd, alpha_bias=alpha_bias,
                              layers=(2, layers[i]), widths=(64, widths[i]), skip1=skips[i])
        nets.append(w.net1)
    return Weights(nets[0], nets[1])


def write_model_dir(path: str, scene: Scene, weights: Weights) -> None:
    """Writes config.ini / dataset_info.txt / model{0,1}.onnx in the exported format
    (minimal 19-key config form of sample_pavillon_16/config.ini; dataset_info.txt as
    src/export.py:47-54)."""
    os.makedirs(path, exist_ok=True)
    sampler = "FromClassifiedDepthAdaptiveNoDepthRange" if scene.use_ndc else "FromClassifiedDepthAdaptive"
    if scene.sampler == "FromClassifiedDepth":
        sampler = "FromClassifiedDepth"
    with open(os.path.join(path, "config.ini"), "w") as f:
        f.write("losses = [%s, MSE]\n" % scene.losses0)
        f.write("posEnc = [nerf, nerf]\n")
        f.write("posEncArgs = [%d-%d, %d-%d]\n" % (scene.pos_enc[0] + scene.pos_enc[1]))
        if scene.sampler == "CoarseFine":
            f.write("inFeatures = [RayMarchFromPoses, RayMarchFromCoarse]\n")
            f.write("outFeatures = [RGBARayMarch, RGBARayMarch]\n")
            f.write("rayMarchSampler = [LinearlySpacedZNearZFar, none]\n")
            if scene.normalization:
                f.write("rayMarchNormalization = [%s, %s]\n" % (scene.normalization, scene.normalization))
            f.write("numRaymarchSamples = [%d, %d]\n" % (scene.num_samples_coarse, scene.num_samples))
        else:
            f.write("inFeatures = [SpherePosDir, RayMarchFromPoses]\n")
            f.write("outFeatures = [Raw, RGBARayMarch]\n")
            f.write("rayMarchSampler = [none, %s]\n" % sampler)
            if scene.normalization:
                f.write("rayMarchNormalization = [InverseSqrtDistCentered, %s]\n" % scene.normalization)
            f.write("numRaymarchSamples = [%d, %d]\n" % (scene.num_samples, scene.num_samples))
        if len(scene.normalization_center) == 3:
            f.write("rayMarchNormalizationCenter = [%r, %r, %r]\n" % tuple(scene.normalization_center))
        f.write("rayMarchSamplingStep = [0.0078125, 0.0078125]\n")
        f.write("rayMarchSamplingNoise = [0.0, 0.0]\n")
        f.write("raySampleInput = [%d, 0]\n" % scene.ray_sample_input)
        f.write("depthTransform = %s\n" % scene.depth_transform)
        f.write("zNear = [%r, %r]\n" % (scene.z_near, scene.z_near))
        f.write("zFar = [%r, %r]\n" % (scene.z_far, scene.z_far))
        f.write("adaptiveSamplingThreshold = %r\n" % scene.threshold)
        f.write("multiDepthFeatures = [%d, %d]\n" % (scene.depth_bins, scene.depth_bins))
        f.write("multiDepthIgnoreValue = [1.01, 1.01]\n")
        f.write("accumulationMult = %s\n" % scene.accumulation_mult)
        f.write("useNDC = %s\n" % ("True" if scene.use_ndc else "False"))
    with open(os.path.join(path, "dataset_info.txt"), "w") as f:
        f.write("view_cell_center = [%r, %r, %r]\n" % tuple(scene.view_cell_center))
        f.write("view_cell_size = [%r, %r, %r]\n" % tuple(scene.view_cell_size))
        f.write("depth_range = [%r, %r]\n" % tuple(scene.depth_range))
        f.write("fov = %r\n" % scene.fov)
        f.write("focal = 0.0\ncamera_scale = 1.0\n")
        f.write("max_depth = %r\n" % scene.max_depth)
    write_onnx_initializers(os.path.join(path, "model0.onnx"), weights.net0)
    write_onnx_initializers(os.path.join(path, "model1.onnx"), weights.net1)


# --------------------------------------------------------------------------------------
# A1  pixel ray table
# --------------------------------------------------------------------------------------

def focal_from_fov(w: int, fov: float) -> float:
    """src/datasets.py:182  focal = .5*w / tan(.5*fov)  (python float64)."""
    return 0.5 * w / math.tan(0.5 * fov)


def generate_ray_directions(w: int, h: int, fov: float, focal: Optional[float] = None,
                            rows: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """src/util/raygeneration.py:10-26, float64 then cast to float32 by the dataset
    (src/datasets.py:190-192).  Row-major [h*w, 3]; ray id = row*w + col.  ``rows`` = (first, end): only those image rows (the same
    element-wise arithmetic, so the values are the full table's; bench.py's CPU baseline renders the frame in slices)."""
    if focal is None:
        focal = focal_from_fov(w, fov)
    x_dist = np.tan(fov / 2) * focal
    y_dist = x_dist * (h / w)
    x_pp = x_dist / (w / 2)
    y_pp = y_dist / (h / 2)
    r0, r1 = rows if rows is not None else (0, h)
    n = r1 - r0
    col = np.arange(w, dtype=np.float64)[None, :].repeat(n, 0)
    row = np.arange(r0, r1, dtype=np.float64)[:, None].repeat(w, 1)
    v = np.empty((n, w, 3), dtype=np.float64)
    v[..., 0] = -(x_dist - x_pp / 2) + x_pp * col
    v[..., 1] = -(y_dist - y_pp / 2) + y_pp * row
    v[..., 2] = focal
    v /= np.linalg.norm(v, axis=2)[..., None]
    v[..., 1] *= -1.0
    v[..., 2] *= -1.0
    return v.reshape(n * w, 3).astype(F32, copy=False)


def camera_rotation(yaw_deg: float, pitch_deg: float) -> np.ndarray:
    """c2w rotation for a z-up world, camera looking along -z with +y up:
    columns = (right, up, -forward).  Direction from yaw/pitch as the viewer's
    Camera::UpdateFeatureRot (adanerf_real_time_viewer/src/camera.cpp:143-158)."""
    y, p = math.radians(yaw_deg), math.radians(pitch_deg)
    fwd = np.array([math.cos(y) * math.cos(p), math.sin(y) * math.cos(p), math.sin(p)])
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    return np.stack([right, up, -fwd], axis=1).astype(F32, copy=False)


# --------------------------------------------------------------------------------------
# A2  world ray + sphere exit + oracle features
# --------------------------------------------------------------------------------------

def positional_encoding(x: np.ndarray, n_freqs: int) -> np.ndarray:
    """src/util/feature_encoding.py:54-73: [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(F-1) x),
    cos(2^(F-1) x)], each a 3-vector; freq bands 2**linspace(0, F-1, F) (exact powers of 2)."""
    x = x.astype(F32)
    parts = [x]
    for k in range(n_freqs):
        f = F32(2.0 ** k)
        parts.append(np.sin(x * f).astype(F32, copy=False))
        parts.append(np.cos(x * f).astype(F32, copy=False))
    return np.concatenate(parts, axis=-1)


def world_rays(dirs_cam: np.ndarray, pose: np.ndarray, rot: np.ndarray, scene: Scene):
    """src/features.py:845-866 (bmm) + compute_ray_offset :769-791.
    Returns (nds [R,3] un-normalised world dirs, p [R,3] sphere-exit points)."""
    rot = rot.astype(F32)
    nds = (dirs_cam.astype(F32) @ rot.T).astype(F32, copy=False)            # R * d per ray
    c = np.array(scene.view_cell_center, dtype=F32)
    o = pose.astype(F32)
    omc = (o - c).astype(F32, copy=False)
    udot = np.sum(omc[None, :] * nds, axis=1, dtype=F32)
    # view_cell_radius is a float64 0-d tensor in the reference; radius**2 promotes the
    # scalar term only (result dtype stays float32 because the other operand is a tensor).
    rad2 = F32(np.float64(scene.radius) ** 2)
    delta = (udot ** 2 - (np.sum(omc ** 2, dtype=F32) - rad2)).astype(F32, copy=False)
    sq = np.sqrt(np.maximum(delta, F32(0))).astype(F32, copy=False)
    dist = (-udot + sq).astype(F32, copy=False)
    p = (o[None, :] + nds * dist[:, None]).astype(F32, copy=False)
    return nds, p


def oracle_features(nds: np.ndarray, p: np.ndarray, scene: Scene) -> np.ndarray:
    """src/features.py:868-888: [PE_dir(nds/||nds||) | PE_pos(p)] and, for raySampleInput = A > 0, the A points
    p + nds * z_a (z_a = to_world of the A bin centres of [0,1]) encoded as PE_pos(x / d1) with the identity part
    scaled back by d1 (d1 = upper end of the warped depth range) -- A * (3 + 6 fp) more columns."""
    fp, fd = scene.pos_enc[0]
    nrm = np.sqrt(np.sum(nds * nds, axis=-1, keepdims=True, dtype=F32)).astype(F32, copy=False)
    cols = [positional_encoding((nds / nrm).astype(F32, copy=False), fd), positional_encoding(p, fp)]
    a = scene.ray_sample_input
    if a:
        zs = ray_sample_depths(scene)
        d1 = F32(scene.depth_range[1])
        pts = (p[:, None, :] + nds[:, None, :] * zs[None, :, None]).astype(F32, copy=False)            # [R, A, 3]
        enc = positional_encoding((pts / d1).astype(F32, copy=False), fp)                               # [R, A, 3 + 6 fp]
        enc[..., :3] = (enc[..., :3] * d1).astype(F32, copy=False)
        cols.append(enc.reshape(pts.shape[0], -1))
    return np.concatenate(cols, axis=-1).astype(F32, copy=False)


def ray_sample_depths(scene: Scene) -> np.ndarray:
    """World depths of the raySampleInput points: to_world(linspace(step/2, 1 - step/2, A)), step = 1/A
    (src/features.py:876-881; always through the depth range, also for NDC scenes)."""
    a = scene.ray_sample_input
    step = 1.0 / a
    t = np.linspace(step / 2, 1.0 - step / 2, a, dtype=F32)
    d0, d1 = scene.depth_range
    if scene.depth_transform == "log":
        return (np.power(F32((d1 - d0) + 1), t).astype(F32, copy=False) - F32(1.0) + F32(d0)).astype(F32, copy=False)
    return (t * F32(d1 - d0) + F32(d0)).astype(F32, copy=False)


# --------------------------------------------------------------------------------------
# A3 / A6  the two MLPs
# --------------------------------------------------------------------------------------

_MATMUL = "numpy"


def set_matmul_backend(name: str) -> None:
    """'numpy' (default; what the golden tests pin) or 'torch': the same fp32 x @ W.T + b through torch's CPU GEMM
    (oneDNN/MKL, all host threads) -- the reference's own CPU path is PyTorch, and numpy's wheel BLAS is ~4x slower
    on these [chunk,256] x [256,256] products, so bench.py's cpu_baseline uses this."""
    global _MATMUL
    if name not in ("numpy", "torch"):
        raise ValueError(name)
    _MATMUL = name


def _linear(x, w, b):
    if _MATMUL == "torch":
        import torch
        with torch.no_grad():
            y = torch.addmm(torch.from_numpy(b), torch.from_numpy(np.ascontiguousarray(x, dtype=F32)), torch.from_numpy(w).t())
        return y.numpy()
    return (x @ w.T + b).astype(F32, copy=False)


def sampling_mlp(x: np.ndarray, net0: Dict[str, np.ndarray]) -> np.ndarray:
    """src/models.py:183-195 (BaseNet.forward, no skips): 7x(Linear+ReLU) + Linear; raw output."""
    n = len([k for k in net0 if k.endswith(".weight")])
    if _MATMUL == "torch":
        return _torch_sampling_mlp(x, net0, n)
    h = x.astype(F32)
    for i in range(n):
        h = _linear(h, net0["layers.%d.weight" % i], net0["layers.%d.bias" % i])
        if i + 1 < n:
            h = np.maximum(h, F32(0))
    return h


def shading_topology(net1: Dict[str, np.ndarray], n_pos: int):
    """(depth D, skip indices) of an exported NeRF trunk: layer i + 1 takes cat([input_pts, h]) iff i is a skip
    (src/models.py:226-228, 257-261), i.e. iff its weight has W + n_pos columns."""
    depth = len([k for k in net1 if k.startswith("pts_linears.") and k.endswith(".weight")])
    width = net1["pts_linears.0.weight"].shape[0]
    skips = [i - 1 for i in range(1, depth) if net1["pts_linears.%d.weight" % i].shape[1] == width + n_pos]
    return depth, skips


def shading_mlp(x: np.ndarray, net1: Dict[str, np.ndarray], n_pos: int = 63) -> np.ndarray:
    """src/models.py:254-277 (NeRF.forward, skips=[4], use_viewdirs=True) -> [rgb(3), alpha(1)] raw."""
    if _MATMUL == "torch":
        return _torch_shading_mlp(x, net1, n_pos)
    x = x.astype(F32)
    pts, views = x[:, :n_pos], x[:, n_pos:]
    h = pts
    depth, skips = shading_topology(net1, n_pos)
    for i in range(depth):
        h = np.maximum(_linear(h, net1["pts_linears.%d.weight" % i], net1["pts_linears.%d.bias" % i]), F32(0))
        if i in skips:
            h = np.concatenate([pts, h], axis=-1)
    alpha = _linear(h, net1["alpha_linear.weight"], net1["alpha_linear.bias"])
    feat = _linear(h, net1["feature_linear.weight"], net1["feature_linear.bias"])
    h = np.concatenate([feat, views], axis=-1)
    h = np.maximum(_linear(h, net1["views_linears.0.weight"], net1["views_linears.0.bias"]), F32(0))
    rgb = _linear(h, net1["rgb_linear.weight"], net1["rgb_linear.bias"])
    return np.concatenate([rgb, alpha], axis=-1).astype(F32, copy=False)


# The two networks with every op in torch (multi-threaded addmm / relu / cat, as the reference's own CPU path runs them):
# used by bench.py's cpu_baseline through set_matmul_backend("torch"); tests/test_oracle_golden.py checks it against the
# numpy path above, which is the one the golden vectors pin.
def _tt(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=F32))


def _torch_sampling_mlp(x, net0, n):
    import torch
    with torch.no_grad():
        h = _tt(x)
        for i in range(n):
            h = torch.addmm(_tt(net0["layers.%d.bias" % i]), h, _tt(net0["layers.%d.weight" % i]).t())
            if i + 1 < n:
                h.relu_()
        return h.numpy()


def _torch_shading_mlp(x, net1, n_pos):
    import torch
    with torch.no_grad():
        x = _tt(x)
        pts, views = x[:, :n_pos], x[:, n_pos:]
        lin = lambda h, nm: torch.addmm(_tt(net1[nm + ".bias"]), h, _tt(net1[nm + ".weight"]).t())
        h = pts
        depth, skips = shading_topology(net1, n_pos)
        for i in range(depth):
            h = lin(h, "pts_linears.%d" % i).relu_()
            if i in skips:
                h = torch.cat([pts, h], -1)
        alpha = lin(h, "alpha_linear")
        feat = lin(h, "feature_linear")
        h = lin(torch.cat([feat, views], -1), "views_linears.0").relu_()
        rgb = lin(h, "rgb_linear")
        return torch.cat([rgb, alpha], -1).numpy()


# --------------------------------------------------------------------------------------
# A4  adaptive selection + compaction
# --------------------------------------------------------------------------------------

def select_adaptive(orc: np.ndarray, n_max: int, thr: float):
    """src/nerf_raymarch_common.py:699-757 (thr > 0 branch), restated as a set rule:
    rank bins by value descending (ties: lower bin first -- the reference's torch.sort is
    unstable there, SURVEY §0, so tie rows at the N-cutoff are implementation-defined and we
    define 'lower bin id wins'); keep the first n_max whose value >= thr; if none is >= thr
    keep the arg-max alone; emit kept bins in ascending depth order.

    Returns (count [R] int32, bins [R,n_max] int16 (-1 padded, ascending),
             weight [R,n_max] float32 (oracle value of the kept bin, 0 padded))."""
    orc = orc.astype(F32)
    r, d = orc.shape
    order = np.argsort(-orc, axis=1, kind="stable")[:, :n_max]
    vals = np.take_along_axis(orc, order, axis=1)
    keep = vals >= F32(thr)
    none = ~keep.any(axis=1)
    keep[none, 0] = True
    count = keep.sum(axis=1).astype(np.int32)
    big = np.where(keep, order, d + 1)
    srt = np.argsort(big, axis=1, kind="stable")
    bins = np.take_along_axis(big, srt, axis=1)
    wts = np.take_along_axis(np.where(keep, vals, F32(0)), srt, axis=1)
    bins = np.where(bins > d, -1, bins).astype(np.int16)
    return count, bins, wts.astype(F32)


def guard_undecided(y: np.ndarray, n_max: int, thr: float, eps: float, transform: str = "none", eps_pair: float = 0.0) -> np.ndarray:
    """Guard band of the two-precision selection (no reference counterpart: the build's own ADANERF_SAMPLING_GUARDED,
    adanerf_amd/csrc/k_select_pair.hip.hpp pair_select; restated here so the CPU suite can check the rule itself).
    ``y`` [R,128]: the sampler's transformed values computed by the cheaper engine, whose RAW outputs are within ``eps``
    of the exact engine's, and -- untransformed outputs only, ``eps_pair`` > 0 -- whose error on a DIFFERENCE between a kept
    value and a not-kept candidate is at most ``eps_pair`` (<= 2 eps; 0 selects 2 eps, which the first bound implies).
    Returns a bool [R]: False only where select_adaptive(x) == select_adaptive(y) for EVERY x that satisfies both bounds.
    Rule (e, ep = the bounds carried through the transform): undecided iff one of the n_max largest values lies within e of thr, or
    more values reach u = max(v_n - ep, thr - e) (arg-max fallback: v_1 - ep) than reach the cut value, or there is a tie at the
    cut / a non-finite value."""
    y = y.astype(F32)
    srt = -np.sort(-y, axis=1)
    c0 = srt[:, 0]
    if transform == "softmax":
        e = (F32(1.01) * F32(np.expm1(2.0 * eps)) * c0).astype(F32, copy=False)
    else:
        e = np.full(len(y), F32(0.25 * eps if transform == "sigmoid" else eps), dtype=F32)
    ep = F32(2) * e
    if transform == "none" and 0.0 < eps_pair <= 2.0 * eps:
        ep = np.full(len(y), F32(eps_pair), dtype=F32)
    top = srt[:, :n_max]
    tn = top[:, n_max - 1]
    none = c0 < F32(thr)
    t = np.where(none, c0, np.maximum(tn, F32(thr)))
    n_eff = np.where(none, 1, n_max)
    total = (y >= t[:, None]).sum(axis=1)
    und = total > n_eff
    und |= (np.abs(top - F32(thr)) <= e[:, None]).any(axis=1)
    u = np.where(none, c0 - ep, np.maximum(tn - ep, F32(thr) - e))
    und |= (y >= u[:, None]).sum(axis=1) != total
    und |= ~np.isfinite(y).all(axis=1)
    return und


def guard_pair_error(y: np.ndarray, x: np.ndarray, n_max: int, thr: float, eps: float) -> np.ndarray:
    """The quantity ``eps_pair`` bounds, per row (pair_monitor / guard_stats_kernel measure it on the device): the largest
    (y_i - x_i) - (y_j - x_j) over bins i kept by the selection from y and candidates j that are not -- values from
    max(v_n - 2 eps, thr - eps) (arg-max fallback: v_1 - 2 eps) up to the cut value; 0 where a row has no such pair.  The sign
    that lets j overtake i: x_j >= x_i needs this to reach y_i - y_j."""
    y = y.astype(F32)
    d = (y - x.astype(F32)).astype(F32, copy=False)
    srt = -np.sort(-y, axis=1)
    c0, tn = srt[:, 0], srt[:, n_max - 1]
    none = c0 < F32(thr)
    t = np.where(none, c0, np.maximum(tn, F32(thr)))
    cut = np.where(none, c0 - F32(2 * eps), np.maximum(tn - F32(2 * eps), F32(thr) - F32(eps)))
    kept = y >= t[:, None]
    cand = (y >= cut[:, None]) & ~kept
    dk = np.where(kept, d, -np.inf).max(axis=1)
    dn = np.where(cand, d, np.inf).min(axis=1)
    pr = dk - dn
    return np.where(np.isfinite(pr) & (pr > 0), pr, 0).astype(F32, copy=False)


def guard_audit_bits(period: int, phase: int, seg: int) -> int:
    """Which of the 32 rays of segment ``seg`` the guarded selection audits at ``phase`` (k_select_pair.hip.hpp audit_bits): bit j set
    iff ((j - phase - seg) & (period - 1)) == 0; period a power of two <= 32, 0 = no audit."""
    if period <= 0:
        return 0
    pattern = 1 if period >= 32 else 0xFFFFFFFF // ((1 << period) - 1)
    return (pattern << ((phase + seg) & (period - 1))) & 0xFFFFFFFF


def guard_refine_list(undecided: np.ndarray, period: int, phase: int, cap_round: int = 0, cycle: int = 0):
    """The list pass 2 of the guarded selection works through (k_compact.hip.hpp refine_list_kernel), restated: ``undecided`` bool [R].
    Returns (rays int32 ascending, audit_only bool) -- every undecided ray, plus the decided rays audited at (period, phase): all of
    them, or with ``cap_round`` > 0 (ADANERF_FLAG_GUARD_AUDIT_FILL) only as many as the last round of ``cap_round`` rays has room for
    beside the undecided ones -- plus one more round where that room is less than a quarter of the candidates -- taken as a window of
    the audit candidates (in ray order) that starts at (cycle * room) mod candidates and wraps round."""
    und = np.asarray(undecided, bool)
    r = np.arange(und.shape[0])
    aud = np.zeros_like(und) if period <= 0 else ((((r & 31) - phase - (r >> 5)) & (period - 1)) == 0) & ~und
    cand = np.flatnonzero(aud)
    if cap_round > 0:
        n_und = int(und.sum())
        rounds = max(1, -(-n_und // cap_round))
        room = max(0, rounds * cap_round - n_und)
        if room < (cand.size + 3) // 4:
            room += cap_round
        room = min(cand.size, room)
        if room < cand.size:
            off = (cycle * room) % cand.size if room > 0 else 0
            keep = ((np.arange(cand.size) - off) % cand.size) < room
            aud = np.zeros_like(und)
            aud[cand[keep]] = True
    take = und | aud
    rays = np.flatnonzero(take).astype(np.int32)
    return rays, aud[rays]


def compact(count: np.ndarray, bins: np.ndarray, wts: np.ndarray):
    """Ray-major, depth-ascending flat order == ``embedded[mapping]`` order of
    src/features.py:438-446, 481-484.  Returns (ray_offset [R] int32 exclusive prefix,
    sample_ray [S] int32, sample_bin [S] int16, sample_w [S] float32)."""
    r, n = bins.shape
    off = np.zeros(r, dtype=np.int64)
    off[1:] = np.cumsum(count.astype(np.int64))[:-1]
    mask = np.arange(n)[None, :] < count[:, None]
    ray = np.repeat(np.arange(r, dtype=np.int32)[:, None], n, 1)[mask]
    return off.astype(np.int32), ray, bins[mask].astype(np.int16), wts[mask].astype(F32)


def bin_t(bins: np.ndarray, n_bins: int = D_BINS) -> np.ndarray:
    """(k + .5) * cell_size with cell_size = 1 / multiDepthFeatures (128 in every shipped config) in float32
    (src/nerf_raymarch_common.py:726-727, 737-741)."""
    return ((bins.astype(F32) + F32(0.5)) * F32(1.0 / n_bins)).astype(F32, copy=False)


def dense_t(scene: Scene, n: int = D_BINS) -> np.ndarray:
    """thr == 0 branch, src/nerf_raymarch_common.py:708-720:
    t = linspace(0,1,n+1)[:-1] + .5/n ; z = near*(1-t) + far*t."""
    t = (np.linspace(0.0, 1.0, n + 1, dtype=F32)[:-1] + F32(0.5 / n)).astype(F32, copy=False)
    return (F32(scene.z_near) * (F32(1.0) - t) + F32(scene.z_far) * t).astype(F32, copy=False)


def to_world_depth(t: np.ndarray, scene: Scene) -> np.ndarray:
    """src/util/depth_transformations.py:37-48 (log) / :57-58 (linear); the adaptive NDC sampler
    (…NoDepthRange, src/nerf_raymarch_common.py:796-851) returns t unchanged, FromClassifiedDepth always maps
    through the depth range (:660), also under NDC."""
    t = t.astype(F32)
    if scene.use_ndc and scene.sampler != "FromClassifiedDepth":
        return t
    d0, d1 = scene.depth_range
    if scene.depth_transform == "log":
        max_v = d1 - d0
        return (np.power(F32(max_v + 1), t).astype(F32, copy=False) - F32(1.0) + F32(d0)).astype(F32, copy=False)
    return (t * F32(d1 - d0) + F32(d0)).astype(F32, copy=False)


def from_world_depth(z: np.ndarray, scene: Scene) -> np.ndarray:
    """Inverse of to_world_depth: LogTransform.from_world (src/util/depth_transformations.py:13-35: values <= 0 after
    subtracting the range minimum become 0.001) / LinearTransform.from_world (:52-54)."""
    z = z.astype(F32)
    d0, d1 = scene.depth_range
    if scene.depth_transform == "log":
        d = (z - F32(d0)).astype(F32, copy=False)
        d = np.where(d <= 0, F32(0.001), d)
        return (np.log(d + F32(1.0)) / F32(math.log((d1 - d0) + 1))).astype(F32, copy=False)
    return ((z - F32(d0)) / F32(d1 - d0)).astype(F32, copy=False)


def oracle_transform(orc: np.ndarray, losses0: str) -> np.ndarray:
    """What the samplers apply to the sampling network's raw outputs before they look at them
    (src/nerf_raymarch_common.py:624-630 / :686-690 / :782-788): sigmoid under BCEWithLogitsLoss, softmax over the
    bins under CrossEntropyLoss[Weighted] (the weighted variant slices to the first `disc` = 128 outputs: all of
    them), nothing otherwise."""
    orc = orc.astype(F32)
    if losses0 == "BCEWithLogitsLoss":
        return sigmoid(orc)
    if losses0 in ("CrossEntropyLoss", "CrossEntropyLossWeighted"):
        e = np.exp(orc - np.max(orc, axis=-1, keepdims=True), dtype=F32)
        return (e / np.sum(e, axis=-1, keepdims=True, dtype=F32)).astype(F32, copy=False)
    return orc


def sample_pdf(orc: np.ndarray, n: int, losses0: str = "BCEWithLogitsLoss") -> np.ndarray:
    """DONeRF sampler: FromClassifiedDepth.generate, src/nerf_raymarch_common.py:606-660, with
    the transform losses[0] selects (oracle_transform; DONeRF trains with BCEWithLogitsLoss -> sigmoid), then
    nerf_sample_pdf (:160-192) with det=True over the 129 bin edges linspace(0,1,129), n+2 uniform u values, first
    and last dropped.  Returns warped depths t [R,n]."""
    w = (oracle_transform(orc, losses0) + F32(1e-5)).astype(F32, copy=False)
    pdf = (w / np.sum(w, axis=-1, keepdims=True, dtype=F32)).astype(F32, copy=False)
    cdf = np.cumsum(pdf, axis=-1, dtype=F32)
    cdf = np.concatenate([np.zeros_like(cdf[:, :1]), cdf], axis=-1)          # [R,129]
    bins = np.linspace(0.0, 1.0, orc.shape[1] + 1, dtype=F32)
    u = np.linspace(0.0, 1.0, n + 2, dtype=F32)
    out = np.empty((orc.shape[0], n + 2), dtype=F32)
    for k in range(n + 2):
        inds = np.sum(cdf <= u[k], axis=1)                                     # searchsorted(..., right=True)
        below = np.maximum(inds - 1, 0)
        above = np.minimum(inds, cdf.shape[1] - 1)
        c0 = np.take_along_axis(cdf, below[:, None], 1)[:, 0]
        c1 = np.take_along_axis(cdf, above[:, None], 1)[:, 0]
        denom = (c1 - c0).astype(F32, copy=False)
        denom = np.where(denom < F32(1e-5), F32(1.0), denom)
        t = ((u[k] - c0) / denom).astype(F32, copy=False)
        out[:, k] = (bins[below] + t * (bins[above] - bins[below])).astype(F32, copy=False)
    return out[:, 1:-1]


def coarse_depths(scene: Scene) -> np.ndarray:
    """LinearlySpacedZNearZFar.generate(det) (src/nerf_raymarch_common.py:310-325): t = linspace(0,1,N+1)[:-1] + 0.5/N,
    near (1-t) + far t with zNear/zFar of net 0, then depth_transform.to_world over the depth range -> [N] world depths."""
    n = scene.num_samples_coarse
    t = (np.linspace(0.0, 1.0, n + 1, dtype=F32)[:-1] + F32(0.5 / n)).astype(F32, copy=False)
    zw = (F32(scene.z_near) * (F32(1.0) - t) + F32(scene.z_far) * t).astype(F32, copy=False)
    return to_world_depth(zw, scene)


def classic_weights(raw: np.ndarray, z: np.ndarray, rays_d: np.ndarray) -> np.ndarray:
    """The `weights` output of nerf_raw2outputs (src/nerf_raymarch_common.py:33-52): alpha_k T_k, [R,N]."""
    raw = raw.astype(F32)
    dists = np.concatenate([z[:, 1:] - z[:, :-1], np.full((z.shape[0], 1), 1e10, dtype=F32)], -1).astype(F32, copy=False)
    dists = (dists * np.sqrt(np.sum(rays_d * rays_d, -1, keepdims=True, dtype=F32))).astype(F32, copy=False)
    alpha = (F32(1.0) - np.exp(-np.maximum(raw[..., 3], F32(0)) * dists, dtype=F32)).astype(F32, copy=False)
    # torch.cumprod / cumsum on CPU accumulate in double (acc_type<float, false>) and store float
    trans = np.cumprod(np.concatenate([np.ones((alpha.shape[0], 1), np.float64), (F32(1.0) - alpha + F32(1e-10)).astype(np.float64)], -1),
                       -1)[:, :-1].astype(F32)
    return (alpha * trans).astype(F32, copy=False)


def sample_pdf_bins(bins: np.ndarray, weights: np.ndarray, n: int) -> np.ndarray:
    """nerf_sample_pdf(bins, weights, n, det=True) (src/nerf_raymarch_common.py:160-192) as RayMarchFromCoarse.batch calls it
    (src/features.py:653-654): bins [R,B] edges, weights [R,B-1]; u = linspace(0,1,n); -> [R,n] depths (ascending)."""
    w = (weights.astype(F32) + F32(1e-5)).astype(F32, copy=False)
    pdf = (w / np.sum(w, -1, keepdims=True, dtype=F32)).astype(F32, copy=False)
    cdf = np.concatenate([np.zeros((pdf.shape[0], 1), F32), np.cumsum(pdf.astype(np.float64), -1).astype(F32, copy=False)], -1)
    u = np.linspace(0.0, 1.0, n, dtype=F32)
    out = np.empty((cdf.shape[0], n), F32)
    nb = cdf.shape[1]
    for r in range(cdf.shape[0]):
        inds = np.searchsorted(cdf[r], u, side="right")
        below = np.maximum(inds - 1, 0)
        above = np.minimum(inds, nb - 1)
        c0, c1 = cdf[r][below], cdf[r][above]
        denom = (c1 - c0).astype(F32, copy=False)
        denom = np.where(denom < F32(1e-5), F32(1.0), denom).astype(F32, copy=False)
        t = ((u - c0) / denom).astype(F32, copy=False)
        b0, b1 = bins[r][below], bins[r][above]
        out[r] = (b0 + t * (b1 - b0)).astype(F32, copy=False)
    return out


def render_coarse_fine(dirs_cam: np.ndarray, pose: np.ndarray, rot: np.ndarray, scene: Scene, weights: Weights,
                       chunk: int = 4096, keep: bool = False, w: int = 0, h: int = 0):
    """Vanilla NeRF with hierarchical sampling: RayMarchFromPoses over LinearlySpacedZNearZFar depths ->
    net 0 -> nerf_raw2outputs weights -> RayMarchFromCoarse (src/features.py:640-672: pdf over the interval mid-points
    with weights[1:-1], merged and sorted with the coarse depths) -> net 1 -> nerf_raw2outputs.  The reference's own
    RayMarchFromCoarse.postprocess cannot run (it unpacks five of nerf_raw2outputs' six values, src/features.py:688); its
    arithmetic is the nerf_raw2outputs call it makes, which is what is restated (and pinned) here."""
    assert scene.sampler == "CoarseFine"
    nc, nf = scene.num_samples_coarse, scene.num_samples
    zc1 = coarse_depths(scene)
    acc: Dict[str, list] = {}
    sc0 = dataclasses.replace(scene, pos_enc=(scene.pos_enc[0], scene.pos_enc[0]))      # shading_inputs reads pos_enc[1]
    for s in range(0, dirs_cam.shape[0], chunk):
        nds, _ = world_rays(dirs_cam[s:s + chunk], pose, rot, scene)
        r = nds.shape[0]
        p = np.repeat(pose.astype(F32)[None], r, 0)      # the rays start at the camera (src/features.py:426-428), not on the view-cell sphere
        zc = np.repeat(zc1[None], r, 0)
        sray = np.repeat(np.arange(r, dtype=np.int32), nc)
        f0 = shading_inputs(p, nds, sray, zc.reshape(-1), sc0, w, h)
        raw0 = shading_mlp(f0, weights.net0, 3 + 6 * scene.pos_enc[0][0])
        rd = nds
        if scene.use_ndc:      # src/features.py:429-431: everything downstream sees the NDC ray (un-normalised direction in the distances)
            p, rd = ndc_rays(h, w, focal_from_fov(w, scene.fov), 1.0, p, nds)
        w0 = classic_weights(raw0.reshape(r, nc, 4), zc, rd)
        rgb0 = composite_classic(raw0.reshape(r, nc, 4), zc, rd)
        mid = (F32(0.5) * (zc[:, 1:] + zc[:, :-1])).astype(F32, copy=False)
        zf = sample_pdf_bins(mid, w0[:, 1:-1], nf)
        za = np.sort(np.concatenate([zc, zf], -1), -1).astype(F32, copy=False)
        sray = np.repeat(np.arange(r, dtype=np.int32), nc + nf)
        p_w = np.repeat(pose.astype(F32)[None], r, 0)
        f1 = shading_inputs(p_w, nds, sray, za.reshape(-1), scene, w, h, unit_dir=False)
        raw1 = shading_mlp(f1, weights.net1, 3 + 6 * scene.pos_enc[1][0])
        rgb, dm, am = composite_classic(raw1.reshape(r, nc + nf, 4), za, rd, aux=True)
        item = dict(rgb=rgb, depth_map=dm, acc_map=am, count=np.full(r, nc + nf, np.int32))
        if keep:
            item.update(nds=rd, p=p, z_coarse=zc, raw0=raw0, weights0=w0, rgb_coarse=rgb0, z_fine=zf, z=za.reshape(-1), feat1=f1, raw=raw1)
        for k, v in item.items():
            acc.setdefault(k, []).append(v)
    return {k: np.concatenate(v) for k, v in acc.items()}


def composite_classic(raw: np.ndarray, z: np.ndarray, rays_d: np.ndarray, aux: bool = False):
    """nerf_raw2outputs, src/nerf_raymarch_common.py:19-68 (no noise, no white background, no oracle weights):
    alpha = 1 - exp(-relu(raw_a) * dist * |d|), dist = z[k+1] - z[k] (last 1e10), rgb = sigmoid(raw).
    raw [R,N,4], z [R,N] world depths, rays_d [R,3] -> [R,3]; aux: (rgb, depth_map = sum w*z, acc_map = sum w)."""
    raw = raw.astype(F32)
    dists = np.concatenate([z[:, 1:] - z[:, :-1], np.full((z.shape[0], 1), 1e10, dtype=F32)], -1).astype(F32, copy=False)
    dists = (dists * np.sqrt(np.sum(rays_d * rays_d, -1, keepdims=True, dtype=F32))).astype(F32, copy=False)
    rgb = sigmoid(raw[..., :3])
    alpha = (F32(1.0) - np.exp(-np.maximum(raw[..., 3], F32(0)) * dists, dtype=F32)).astype(F32, copy=False)
    trans = np.cumprod(np.concatenate([np.ones((alpha.shape[0], 1), F32), F32(1.0) - alpha + F32(1e-10)], -1), -1, dtype=F32)[:, :-1]
    wts = (alpha * trans).astype(F32, copy=False)
    out = np.sum(wts[..., None] * rgb, -2, dtype=F32).astype(F32, copy=False)
    if aux:
        return out, np.sum(wts * z, -1, dtype=F32).astype(F32, copy=False), np.sum(wts, -1, dtype=F32).astype(F32, copy=False)
    return out


# --------------------------------------------------------------------------------------
# A5  sample position + normalisation + shading PE
# --------------------------------------------------------------------------------------

def ndc_rays(h: int, w: int, focal: float, near: float, rays_o: np.ndarray, rays_d: np.ndarray):
    """src/nerf_raymarch_common.py:71-88."""
    rays_o = rays_o.astype(F32)
    rays_d = rays_d.astype(F32)
    t = (-(F32(near) + rays_o[:, 2]) / rays_d[:, 2]).astype(F32, copy=False)
    rays_o = (rays_o + t[:, None] * rays_d).astype(F32, copy=False)
    sw = F32(-1.0 / (w / (2.0 * focal)))
    sh = F32(-1.0 / (h / (2.0 * focal)))
    o0 = sw * rays_o[:, 0] / rays_o[:, 2]
    o1 = sh * rays_o[:, 1] / rays_o[:, 2]
    o2 = F32(1.0) + F32(2.0 * near) / rays_o[:, 2]
    d0 = sw * (rays_d[:, 0] / rays_d[:, 2] - rays_o[:, 0] / rays_o[:, 2])
    d1 = sh * (rays_d[:, 1] / rays_d[:, 2] - rays_o[:, 1] / rays_o[:, 2])
    d2 = F32(-2.0 * near) / rays_o[:, 2]
    return (np.stack([o0, o1, o2], -1).astype(F32, copy=False), np.stack([d0, d1, d2], -1).astype(F32, copy=False))


def normalize_positions(x: np.ndarray, scene: Scene) -> np.ndarray:
    """nerf_get_normalization_function(rayMarchNormalization[net])(x, centre, max_depth), src/nerf_raymarch_common.py:195-244; centre =
    rayMarchNormalizationCenter when the config holds three values, else view_cell_center (src/features.py:460-467); a config
    without the key (normalization == "") gets normalization_max_depth (src/features.py:319-324)."""
    name = scene.normalization or "MaxDepth"
    md = F32(scene.max_depth)
    if name == "None":
        return x
    if name == "MaxDepth":
        return (x / md).astype(F32, copy=False)
    c = np.array(scene.normalization_center if len(scene.normalization_center) == 3 else scene.view_cell_center, dtype=F32)
    loc = (x - c).astype(F32, copy=False)
    if name == "Centered":
        return loc
    if name == "MaxDepthCentered":
        return (loc / md).astype(F32, copy=False)
    n = np.sqrt(np.sum(loc * loc, -1, dtype=F32)).astype(F32, copy=False)
    if name == "InverseSqrtDistCentered":      # :226-230
        local = np.sqrt(n).astype(F32, copy=False)
        return (loc / (F32(math.sqrt(scene.max_depth)) * local[:, None])).astype(F32, copy=False)
    if name == "InverseDistCentered":          # :219-223
        return (loc * (F32(1.0) - F32(1.0) / (F32(1.0) + n))[:, None]).astype(F32, copy=False)
    if name == "LogCentered":                  # :211-216; LogTransform.from_world clamps its argument IN PLACE (util/depth_transformations.py:21-27)
        n = np.where(n <= 0, F32(0.001), n).astype(F32, copy=False)
        lt = (np.log(n + F32(1.0), dtype=F32) / F32(math.log(scene.max_depth + 1.0))).astype(F32, copy=False)
        return (loc * (lt / n)[:, None]).astype(F32, copy=False)
    raise NotImplementedError(name)


def shading_inputs(p: np.ndarray, nds: np.ndarray, sample_ray: np.ndarray, z: np.ndarray,
                   scene: Scene, w: int = 0, h: int = 0, unit_dir: bool = True, position_ulp: Optional[np.ndarray] = None) -> np.ndarray:
    """src/features.py:420-479: x = o + d*z; normalise; [PE_pos(x^) | PE_dir(dir)].
    ``z`` is the per-sample world depth (to_world_depth of the bin's t).  ``position_ulp`` ([S,3], not part of the reference): the
    normalised positions moved by that many ulp before the encoding -- the conditioning analysis of tests/fuzz_parity.py (two correct fp32
    evaluations of o + d z and of the normalisation differ in their last bits, and an encoding with F bands multiplies that by 2^(F-1))."""
    fp, fd = scene.pos_enc[1]
    o, d = p, nds
    dir_pe = nds
    if scene.use_ndc:
        o, d = ndc_rays(h, w, focal_from_fov(w, scene.fov), 1.0, p, nds)
        # RayMarchFromPoses encodes the normalised NDC direction (src/features.py:431); RayMarchFromCoarse is handed rays_d and
        # encodes it as it is (src/features.py:654-668): unit_dir = False
        dir_pe = (d / np.sqrt(np.sum(d * d, -1, keepdims=True, dtype=F32))).astype(F32, copy=False) if unit_dir else d
    x = (o[sample_ray] + d[sample_ray] * z[:, None]).astype(F32, copy=False)
    x = normalize_positions(x, scene)
    if position_ulp is not None:
        x = (x + position_ulp.astype(F32) * np.spacing(np.abs(x).astype(F32))).astype(F32, copy=False)
    return np.concatenate([positional_encoding(x, fp),
                           positional_encoding(dir_pe[sample_ray], fd)], -1).astype(F32, copy=False)


# --------------------------------------------------------------------------------------
# A7  compositing
# --------------------------------------------------------------------------------------

def sigmoid(x):
    x = x.astype(F32)
    return (F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32))).astype(F32, copy=False)


def composite(raw: np.ndarray, sample_w: np.ndarray, ray_offset: np.ndarray, count: np.ndarray,
              accumulation_mult: str = "alpha", z: Optional[np.ndarray] = None):
    """src/nerf_raymarch_common.py:91-144 (adaptive_raw2outputs): sigmoid on all four
    channels, alpha *= oracle value, w = alpha * cumprod(1 - alpha + 1e-10) (exclusive),
    rgb = sum w*c.  Inactive slots contribute alpha 0.  fp32, unclamped.  [R,3].
    With ``z`` (world depth per sample): also depth_map = sum w*z and acc_map = sum w (:137-139) -> (rgb, depth, acc)."""
    r = count.shape[0]
    n = int(count.max()) if r else 0
    sg = sigmoid(raw)
    rgb = np.zeros((r, 3), dtype=F32)
    trans = np.ones(r, dtype=F32)
    depth = np.zeros(r, dtype=F32)
    acc = np.zeros(r, dtype=F32)
    for s in range(n):
        act = count > s
        idx = np.where(act, ray_offset + s, 0)
        a = np.where(act, sg[idx, 3], F32(0)).astype(F32, copy=False)
        if accumulation_mult == "alpha":
            a = (a * np.where(act, sample_w[idx], F32(0))).astype(F32, copy=False)
        wgt = (a * trans).astype(F32, copy=False)
        if accumulation_mult == "weights":
            wgt = (wgt * np.where(act, sample_w[idx], F32(0))).astype(F32, copy=False)
        rgb += (wgt[:, None] * np.where(act[:, None], sg[idx, :3], F32(0))).astype(F32, copy=False)
        if z is not None:
            depth += (wgt * np.where(act, z[idx], F32(0))).astype(F32, copy=False)
            acc += wgt
        trans = (trans * (F32(1.0) - a + F32(1e-10))).astype(F32, copy=False)
    return rgb if z is None else (rgb, depth, acc)


def effective_mult(scene: "Scene") -> str:
    """The kept oracle values reach compositing only under losses[0] == NeRFWeightMultiplicationLoss
    (src/features.py:503-505 puts them into the feature dict, :517-519 hands them on as ``depth``); otherwise
    adaptive_raw2outputs sees depth=None and accumulationMult has no effect (src/nerf_raymarch_common.py:123-133)."""
    return scene.accumulation_mult if scene.losses0 == "NeRFWeightMultiplicationLoss" else ""


def oracle_view(orc: np.ndarray) -> np.ndarray:
    """Sampling-network debug view, [n,128] raw outputs -> [n,4] uint8.  Restates the VIEWER kernel samplesToImage
    (adanerf_real_time_viewer/src/cuda/base_cuda_kernels.cu:487-528; the PyTorch path has no counterpart, and the viewer
    cannot be built here, so this function is pinned only by reading the kernel: "parity unpinned").  Stable descending
    sort of the 128 values, first three bin ids -> (0.5 + id) / 128 in R, G, B, clamp, * 255, truncate."""
    order = np.argsort(-orc.astype(F32), axis=1, kind="stable")[:, :3]
    v = np.clip((F32(0.5) + order.astype(F32)) / F32(128.0), F32(0), F32(1)) * F32(255.0)
    out = np.full((orc.shape[0], 4), 255, dtype=np.uint8)
    out[:, :3] = v.astype(np.uint8)
    return out


def to_rgba8(rgb: np.ndarray) -> np.ndarray:
    """Viewer output contract (adaptive_cuda_kernels.cu:846-851): (uchar)(clamp(v,0,1)*255), A=255."""
    v = np.clip(rgb.astype(F32), F32(0), F32(1)) * F32(255.0)
    out = np.full((rgb.shape[0], 4), 255, dtype=np.uint8)
    out[:, :3] = v.astype(np.uint8)
    return out


def psnr(a: np.ndarray, b: np.ndarray) -> float:
    """src/evaluate.py:49-54: 10*log10(1/mse), mse over all values."""
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * math.log10(1.0 / mse)


# --------------------------------------------------------------------------------------
# A8  frame / chunk driver
# --------------------------------------------------------------------------------------

def render_rays(dirs_cam: np.ndarray, pose: np.ndarray, rot: np.ndarray, scene: Scene,
                weights: Weights, w: int = 0, h: int = 0, chunk: int = 8192,
                keep: bool = False):
    """TrainConfig.inference (src/train_data.py:278-299) over chunks
    (src/evaluate.py:206-241).  Returns dict with rgb [R,3], count [R] and, when ``keep``,
    every intermediate the golden fixtures hold."""
    if scene.sampler == "CoarseFine":
        return render_coarse_fine(dirs_cam, pose, rot, scene, weights, min(chunk, 4096), keep, w, h)
    out_rgb = []
    out_cnt = []
    out_depth = []
    out_acc = []
    kept: Dict[str, list] = {}
    n_pos = 3 + 6 * scene.pos_enc[1][0]
    for s in range(0, dirs_cam.shape[0], chunk):
        dc = dirs_cam[s:s + chunk]
        nds, p = world_rays(dc, pose, rot, scene)
        feat0 = oracle_features(nds, p, scene)
        orc = sampling_mlp(feat0, weights.net0)
        if scene.sampler == "FromClassifiedDepth":
            r, n = orc.shape[0], scene.num_samples
            tt = sample_pdf(orc, n, scene.losses0)
            z2 = to_world_depth(tt, scene)
            sray = np.repeat(np.arange(r, dtype=np.int32), n)
            feat1 = shading_inputs(p, nds, sray, z2.reshape(-1), scene, w, h)
            raw = shading_mlp(feat1, weights.net1, n_pos)
            # |rays_d| of the rays the samples were placed on: the NDC directions under useNDC (src/features.py:426-431, 507)
            rd = ndc_rays(h, w, focal_from_fov(w, scene.fov), 1.0, p, nds)[1] if scene.use_ndc else nds
            rgb, dm, am = composite_classic(raw.reshape(r, n, 4), z2, rd, aux=True)
            out_rgb.append(rgb)
            out_depth.append(dm)
            out_acc.append(am)
            out_cnt.append(np.full(r, n, dtype=np.int32))
            if keep:
                for k, v in dict(nds=nds, p=p, feat0=feat0, orc=orc, z=z2.reshape(-1), t=tt, feat1=feat1, raw=raw).items():
                    kept.setdefault(k, []).append(v)
            continue
        orc_t = oracle_transform(orc, scene.losses0)      # what the sampler thresholds / ranks and hands on as weights
        if scene.threshold == 0.0:
            r = orc.shape[0]
            count = np.full(r, D_BINS, dtype=np.int32)
            bins = np.repeat(np.arange(D_BINS, dtype=np.int16)[None], r, 0)
            wts = orc_t
            tt = np.repeat(dense_t(scene)[None], r, 0)
        else:
            count, bins, wts = select_adaptive(orc_t, scene.num_samples, scene.threshold)
            tt = bin_t(bins, scene.depth_bins)
        off, sray, sbin, sw = compact(count, bins, wts)
        mask = np.arange(bins.shape[1])[None, :] < count[:, None]
        z = to_world_depth(tt[mask], scene)
        feat1 = shading_inputs(p, nds, sray, z, scene, w, h)
        raw = shading_mlp(feat1, weights.net1, n_pos)
        rgb, dm, am = composite(raw, sw, off, count, effective_mult(scene), z)
        out_rgb.append(rgb)
        out_depth.append(dm)
        out_acc.append(am)
        out_cnt.append(count)
        if keep:
            for k, v in dict(nds=nds, p=p, feat0=feat0, orc=orc, bins=bins, wts=wts, z=z,
                             feat1=feat1, raw=raw).items():
                kept.setdefault(k, []).append(v)
    res = {"rgb": np.concatenate(out_rgb), "count": np.concatenate(out_cnt),
           "depth_map": np.concatenate(out_depth), "acc_map": np.concatenate(out_acc)}
    if keep:
        for k, v in kept.items():
            res[k] = np.concatenate(v)
    return res


def render_frame(scene: Scene, weights: Weights, w: int, h: int, pose: np.ndarray, rot: np.ndarray,
                 chunk: int = 8192, rows: Optional[Tuple[int, int]] = None):
    dirs = generate_ray_directions(w, h, scene.fov, rows=rows)
    return render_rays(dirs, pose, rot, scene, weights, w, h, chunk)
