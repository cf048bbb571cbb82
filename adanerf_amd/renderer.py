"""ctypes host mirror of the reference viewer's object model over the C ABI (include/adanerf_hip.h).

Names follow adanerf_real_time_viewer: ``Settings`` (include/settings.h:12-34, CLI semantics of
src/settings.cpp:15-47) and ``NeuralRenderer`` with ``init()`` / ``render()``
(include/neuralrenderer.h:53-55).  Stage-level methods mirror the reference launchers
(include/cuda/adanerf_cuda_kernels.cuh:20-74) and exist for the parity tests.

The product path is the HIP library.  Nothing here falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .build import library_path

PREC_BF16, PREC_FP16, PREC_FP32 = 0, 1, 2
_PREC = {"bf16": PREC_BF16, "fp16": PREC_FP16, "f16": PREC_FP16, "fp32": PREC_FP32, "f32": PREC_FP32}

BUF_RAYS, BUF_ORACLE, BUF_RAY_OFFSETS, BUF_RAY_COUNTS, BUF_SAMPLE_KEY, BUF_SAMPLE_W, BUF_RAW, BUF_TOTAL, BUF_SAMPLE_Z, BUF_RAW_COARSE = range(10)
SAMPLER_ADAPTIVE, SAMPLER_PDF, SAMPLER_COARSE_FINE = 0, 1, 2
FLAG_KEEP_ORACLE, FLAG_WAVE_SELECT, FLAG_NO_GUARD_CACHE, FLAG_GUARD_AUDIT_FILL = 1, 2, 4, 8
ABI_VERSION = 4
GUARD_FROM = {0: "none", 1: "options", 2: "record", 3: "calibration", 4: "monitor"}
SAMPLING_MODES = {"split": 0, "fp16x3": 0, "fp32": 1, "fp16": 2, "guarded": 3}
# The default rule: the sampling mode that is exact by construction (split) unless the guarded mode is at least this much faster ON THE
# WORKLOAD AT HAND -- choose_sampling() measures it; `sampling="auto"` / `adanerf --sampling auto` apply it.
DEFAULT_RULE_MARGIN = 0.08


class AdaNeRFError(RuntimeError):
    pass


class _Options(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("batch_rays", C.c_int32), ("device_id", C.c_int32),
                ("precision", C.c_int32), ("num_samples", C.c_int32), ("threshold", C.c_float),
                ("shard_rank", C.c_int32), ("shard_world", C.c_int32), ("strip_rows", C.c_int32),
                ("sampling_mode", C.c_int32), ("flags", C.c_int32), ("guard_eps", C.c_float), ("guard_eps_pair", C.c_float),
                ("guard_audit_period", C.c_int32), ("reserved", C.c_int32 * 1)]


class Info(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("rays_local", C.c_int32),
                ("rays_local_max", C.c_int32), ("batch_rays", C.c_int32), ("n_in0", C.c_int32), ("n_in1", C.c_int32),
                ("num_samples", C.c_int32), ("threshold", C.c_float), ("dense", C.c_int32), ("use_ndc", C.c_int32),
                ("precision", C.c_int32), ("compute_units", C.c_int32), ("fov", C.c_float), ("focal", C.c_float),
                ("view_cell_center", C.c_float * 3), ("view_cell_radius", C.c_float), ("depth_range", C.c_float * 2),
                ("max_depth", C.c_float), ("sampler_mode", C.c_int32), ("view_cell_size", C.c_float * 3), ("num_samples_coarse", C.c_int32), ("guard_eps", C.c_float),
                ("guard_eps_pair", C.c_float), ("guard_audit_period", C.c_int32), ("guard_calib_source", C.c_int32), ("guard_calib_poses", C.c_int32),
                ("reserved", C.c_int32 * 8)]


class Stats(C.Structure):
    _fields_ = [("total_samples", C.c_int64), ("rays", C.c_int32), ("batches", C.c_int32), ("ms_total", C.c_float),
                ("ms_sample_mlp", C.c_float), ("ms_compact", C.c_float), ("ms_shade_mlp", C.c_float),
                ("ms_composite", C.c_float), ("shade_launches", C.c_int32), ("sample_launches", C.c_int32),
                ("sampling_overflow", C.c_int32), ("rays_refined", C.c_int32), ("guard_max_seen", C.c_float),
                ("guard_violations", C.c_int32), ("guard_widened", C.c_int32), ("guard_pair_seen", C.c_float), ("guard_audited", C.c_int32),
                ("guard_audit_mismatch", C.c_int32), ("reserved", C.c_int32 * 5)]


EXPORTS = ["adanerf_create", "adanerf_destroy", "adanerf_get_info", "adanerf_last_error", "adanerf_abi_version", "adanerf_struct_sizes", "adanerf_set_camera",
           "adanerf_render", "adanerf_set_aux_outputs", "adanerf_set_disp_output", "adanerf_assemble_strips", "adanerf_sync", "adanerf_set_stream", "adanerf_set_profiling",
           "adanerf_collect_stats", "adanerf_ray_features", "adanerf_sample_mlp",
           "adanerf_compact", "adanerf_compact_guarded", "adanerf_calibrate_guard", "adanerf_guard_calibration_file", "adanerf_shade_features", "adanerf_shade_mlp", "adanerf_shade_mlp_z", "adanerf_sample_pdf", "adanerf_sample_uniform", "adanerf_shade_mlp_coarse", "adanerf_sample_from_coarse",
           "adanerf_composite", "adanerf_composite_classic", "adanerf_copy_result_sampling_network",
           "adanerf_render_oracle", "adanerf_gather_to", "adanerf_probe_mfma", "adanerf_malloc",
           "adanerf_free", "adanerf_memcpy_h2d", "adanerf_memcpy_d2h", "adanerf_get_buffer"]

_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libadanerf_hip.so (built in-tree by adanerf_amd.build).  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("ADANERF_LIB") or library_path()   # ADANERF_LIB: tools/ablate.sh variants
    if not os.path.exists(p):
        raise AdaNeRFError("HIP library not built: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % p)
    lib = C.CDLL(p)
    vp, i32, f32p = C.c_void_p, C.c_int32, C.c_void_p
    lib.adanerf_create.argtypes = [C.c_char_p, C.POINTER(_Options), C.POINTER(vp)]
    lib.adanerf_destroy.argtypes = [vp]
    lib.adanerf_get_info.argtypes = [vp, C.POINTER(Info)]
    lib.adanerf_last_error.argtypes = [vp]
    lib.adanerf_last_error.restype = C.c_char_p
    lib.adanerf_set_camera.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.adanerf_render.argtypes = [vp, vp, vp, C.POINTER(Stats)]
    lib.adanerf_assemble_strips.argtypes = [vp, vp, vp]
    lib.adanerf_set_aux_outputs.argtypes = [vp, vp, vp]
    lib.adanerf_set_disp_output.argtypes = [vp, vp]
    lib.adanerf_sync.argtypes = [vp]
    lib.adanerf_set_stream.argtypes = [vp, vp]
    lib.adanerf_set_profiling.argtypes = [vp, i32]
    lib.adanerf_collect_stats.argtypes = [vp, C.POINTER(Stats), C.POINTER(i32)]
    lib.adanerf_ray_features.argtypes = [vp, i32, i32, f32p, f32p]
    lib.adanerf_sample_mlp.argtypes = [vp, i32, i32, f32p, f32p]
    lib.adanerf_compact.argtypes = [vp, vp, i32, i32, C.c_float, vp, vp, vp, vp, vp]
    lib.adanerf_compact_guarded.argtypes = [vp, vp, vp, i32, i32, C.c_float, C.c_float, C.c_float, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.adanerf_calibrate_guard.argtypes = [vp, i32, C.c_uint32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.adanerf_guard_calibration_file.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.adanerf_abi_version.argtypes = []
    lib.adanerf_struct_sizes.argtypes = [C.POINTER(i32)]
    lib.adanerf_shade_features.argtypes = [vp, vp, vp, i32, vp]
    lib.adanerf_shade_mlp.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    lib.adanerf_composite.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp]
    lib.adanerf_shade_mlp_z.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp]
    lib.adanerf_sample_pdf.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.adanerf_sample_uniform.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.adanerf_shade_mlp_coarse.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    lib.adanerf_sample_from_coarse.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.adanerf_composite_classic.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    lib.adanerf_copy_result_sampling_network.argtypes = [vp, vp, i32, vp]
    lib.adanerf_render_oracle.argtypes = [vp, vp]
    lib.adanerf_gather_to.argtypes = [vp, vp, vp, vp, C.c_size_t]
    lib.adanerf_probe_mfma.argtypes = [vp, i32, i32, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.adanerf_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.adanerf_free.argtypes = [vp, vp]
    lib.adanerf_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.adanerf_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    lib.adanerf_get_buffer.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t)]
    for name in EXPORTS:
        if name != "adanerf_last_error":
            getattr(lib, name).restype = C.c_int
    # the handshake include/adanerf_hip.h asks of a binding: structs are written whole, so version and sizes must agree
    sizes = (i32 * 3)()
    lib.adanerf_struct_sizes(sizes)
    if lib.adanerf_abi_version() != ABI_VERSION or list(sizes) != [C.sizeof(_Options), C.sizeof(Info), C.sizeof(Stats)]:
        raise AdaNeRFError("%s: ABI %d with struct sizes %s, this host expects ABI %d with %s" %
                           (p, lib.adanerf_abi_version(), list(sizes), ABI_VERSION, [C.sizeof(_Options), C.sizeof(Info), C.sizeof(Stats)]))
    if path is None:
        _lib = lib
    return lib


@dataclass
class Settings:
    """Viewer CLI settings (adanerf_real_time_viewer/src/settings.cpp:15-47)."""
    model_path: str
    width: int = 800
    height: int = 800
    batch_size: int = -1          # -bs: <= 0 -> width*height, else min(bs, w*h)
    number_of_batches: int = 1    # -nb (overridden by batch_size)
    write_images: bool = False
    is_debug: bool = True         # headless always

    @property
    def total_size(self) -> int:
        return self.width * self.height

    def resolved_batch(self) -> int:
        if self.batch_size and self.batch_size > 0:
            return min(self.batch_size, self.total_size)
        if self.number_of_batches > 1:
            return -(-self.total_size // self.number_of_batches)
        return self.total_size


class DeviceArray:
    """A device allocation owned by the library's allocator, with numpy round trips."""

    def __init__(self, r: "NeuralRenderer", shape, dtype):
        self.r = r
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        r._check(r.lib.adanerf_malloc(r.handle, max(self.nbytes, 1), C.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray) -> "DeviceArray":
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        if self.nbytes:
            self.r._check(self.r.lib.adanerf_memcpy_h2d(self.r.handle, self.ptr, a.ctypes.data, self.nbytes))
        return self

    def numpy(self, count: Optional[int] = None) -> np.ndarray:
        shape = self.shape if count is None else (count,) + self.shape[1:]
        out = np.empty(shape, dtype=self.dtype)
        if out.nbytes:
            self.r._check(self.r.lib.adanerf_memcpy_d2h(self.r.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr and self.r.handle:
            self.r.lib.adanerf_free(self.r.handle, self.ptr)
        self.ptr = None


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return x.ptr
    if hasattr(x, "data_ptr"):      # torch tensor on the context's device
        return x.data_ptr()
    return int(x)


def choose_sampling(settings, pos=None, rot_c2w=None, precision="bf16", frames: int = 6, warmup: int = 2, margin: float = DEFAULT_RULE_MARGIN, **kw):
    """The default rule per workload: renders `frames` frames of this model / frame size / threshold in the split mode (exact by construction) and in
    the guarded mode (same frames while its measured band holds) and returns ("guarded" | "split", record): guarded only if it is >= margin faster
    here.  pos / rot_c2w: the camera to measure with (default: view-cell centre, looking down -z).  A configuration the guarded mode does not
    exist for (N > 16, run-time-shaped sampling networks, fp32) yields "split"."""
    import time
    fps, note = {}, None
    for mode in ("split", "guarded"):
        try:
            with NeuralRenderer(settings, precision=precision, sampling=mode, **kw) as r:
                p = np.asarray(pos if pos is not None else list(r.info.view_cell_center), np.float32)
                m = np.asarray(rot_c2w if rot_c2w is not None else np.eye(3), np.float32)
                r.set_camera(p, m)
                out = r.empty((r.info.rays_local, 4), np.uint8)
                for _ in range(warmup):
                    r.render(out, None)
                r.sync()
                t0 = time.perf_counter()
                for _ in range(frames):
                    r.render(out, None)
                r.sync()
                fps[mode] = frames / (time.perf_counter() - t0)
        except AdaNeRFError as e:
            if mode == "split":
                raise
            note = "guarded mode not available: %s" % e
    ahead = (fps["guarded"] / fps["split"] - 1.0) if "guarded" in fps else None
    choice = "guarded" if (ahead is not None and ahead >= margin) else "split"
    return choice, {"choice": choice, "split_fps": fps.get("split"), "guarded_fps": fps.get("guarded"), "guarded_ahead": ahead, "margin": margin,
                    "frames": frames, "note": note}


class NeuralRenderer:
    """Headless counterpart of the viewer's NeuralRenderer: ``init()`` loads the model directory and
    builds the device state, ``render()`` produces one frame."""

    def __init__(self, settings: Settings, precision="bf16", device_id: int = 0, num_samples: int = 0,
                 threshold: float = -1.0, shard_rank: int = 0, shard_world: int = 1, strip_rows: int = 8,
                 sampling: Optional[str] = None, lib_path: Optional[str] = None, keep_oracle: bool = False, wave_select: bool = False,
                 guard_eps: float = 0.0, guard_eps_pair: float = 0.0, guard_audit_period: int = 0, guard_cache: bool = True,
                 guard_audit_fill: bool = True):
        """sampling: arithmetic of the sampling network -- "split" (default: split-fp16, fp32-accurate on every ray, the selection exact by
        construction), "fp32" (fp32 MFMA), "guarded" (opt-in: plain fp16 for every ray + the split engine where the audited guard band cannot
        decide; the split engine's selections as long as the measured band holds -- monitored, audited, widened when violated), "fp16" (opt-in
        speed mode, the viewer's TensorRT arithmetic; selections differ on ~1 % of rays).  The default follows a rule: the mode that is
        exact by construction, unless the guarded mode is >= 8 % faster on the same box in the same bench.py run -- measured 4-5 %.  Same default in
        the `adanerf` CLI and bench.py.  guard_*: include/adanerf_hip.h adanerf_options.  guard_audit_fill (default): the audit fills the
        refinement pass's last round instead of adding one, and never audits less than a quarter of the 1 / period quota
        (ADANERF_FLAG_GUARD_AUDIT_FILL); False: exactly 1 / period of all rays every frame."""
        if sampling is None:
            sampling = "split"
        self.sampling_choice = None
        if sampling == "auto":      # the default rule applied to this model / frame size / threshold on this box, camera at the view-cell centre
            if precision == "fp32" or _PREC.get(precision, precision) == _PREC["fp32"]:
                sampling = "split"
            else:
                sampling, self.sampling_choice = choose_sampling(settings, precision=precision, device_id=device_id, num_samples=num_samples, threshold=threshold,
                                                                 shard_rank=shard_rank, shard_world=shard_world, strip_rows=strip_rows, lib_path=lib_path)
        self.settings = settings
        self.lib = load_library(lib_path)
        self.handle = None
        self._opt = _Options(width=settings.width, height=settings.height, batch_rays=settings.resolved_batch(),
                             device_id=device_id, precision=_PREC[precision] if isinstance(precision, str) else int(precision),
                             num_samples=num_samples, threshold=threshold, shard_rank=shard_rank,
                             shard_world=shard_world, strip_rows=strip_rows,
                             sampling_mode=SAMPLING_MODES[sampling], guard_eps=guard_eps, guard_eps_pair=guard_eps_pair,
                             guard_audit_period=guard_audit_period,
                             flags=(FLAG_KEEP_ORACLE if keep_oracle else 0) | (FLAG_WAVE_SELECT if wave_select else 0) |
                                   (0 if guard_cache else FLAG_NO_GUARD_CACHE) | (FLAG_GUARD_AUDIT_FILL if guard_audit_fill else 0))
        self.info = Info()
        self.last_stats = Stats()
        self._own = []

    # -- lifecycle -------------------------------------------------------------------------------
    def init(self) -> bool:
        h = C.c_void_p()
        rc = self.lib.adanerf_create(self.settings.model_path.encode(), C.byref(self._opt), C.byref(h))
        if rc != 0:
            raise AdaNeRFError("adanerf_create(%s) failed (%d): %s" %
                               (self.settings.model_path, rc, self.lib.adanerf_last_error(None).decode()))
        self.handle = h.value
        self._check(self.lib.adanerf_get_info(self.handle, C.byref(self.info)))
        return True

    def close(self):
        if self.handle:
            for a in self._own:
                a.free()
            self._own = []
            self.lib.adanerf_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        if not self.handle:
            self.init()
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            msg = self.lib.adanerf_last_error(self.handle).decode() if self.handle else "?"
            raise AdaNeRFError("libadanerf_hip error %d: %s" % (rc, msg))

    def empty(self, shape, dtype) -> DeviceArray:
        a = DeviceArray(self, shape, dtype)
        self._own.append(a)
        return a

    def to_device(self, arr: np.ndarray) -> DeviceArray:
        return self.empty(arr.shape, arr.dtype).upload(arr)

    # -- per frame ---------------------------------------------------------------------------------
    def set_camera(self, pos, rot_c2w):
        p = np.ascontiguousarray(pos, dtype=np.float32).reshape(3)
        r = np.ascontiguousarray(rot_c2w, dtype=np.float32).reshape(9)
        self._check(self.lib.adanerf_set_camera(self.handle, p.ctypes.data_as(C.POINTER(C.c_float)),
                                                r.ctypes.data_as(C.POINTER(C.c_float))))

    def render(self, rgba8_out=None, rgb_out=None, stats: bool = False) -> Optional[Stats]:
        """One frame into caller-owned device buffers ([rays_local] uchar4 / [rays_local,3] fp32).
        ``stats=True`` synchronises and returns the per-stage timing record."""
        st = Stats() if stats else None
        self._check(self.lib.adanerf_render(self.handle, _ptr(rgba8_out), _ptr(rgb_out), C.byref(st) if stats else None))
        if stats:
            self.last_stats = st
        return st

    def set_aux_outputs(self, depth_map=None, acc_map=None):
        """Subsequent renders also fill [rays_local] fp32 depth_map (sum w z) / acc_map (sum w); None switches them off."""
        self._check(self.lib.adanerf_set_aux_outputs(self.handle, _ptr(depth_map), _ptr(acc_map)))

    def set_disp_output(self, disp_map=None):
        """Subsequent renders also fill [rays_local] fp32 disp_map = 1 / max(1e-10, depth_map / acc_map); None switches it off."""
        self._check(self.lib.adanerf_set_disp_output(self.handle, _ptr(disp_map)))

    def gather_from(self, dst, src_renderer: "NeuralRenderer", src, nbytes: int):
        """Stream-ordered copy of ``nbytes`` from ``src`` (on ``src_renderer``'s device / stream) into ``dst`` on this
        renderer's device; this renderer's stream waits for it (single-process multi-GPU strip exchange)."""
        self._check(self.lib.adanerf_gather_to(self.handle, _ptr(dst), src_renderer.handle, _ptr(src), nbytes))

    def render_oracle(self, rgba8_out):
        """Sampling-network debug view of this rank's rays (the viewer's 'O' key): [rays_local] uchar4."""
        self._check(self.lib.adanerf_render_oracle(self.handle, _ptr(rgba8_out)))

    def copy_result_sampling_network(self, oracle, n_rays: int, rgba8_out):
        self._check(self.lib.adanerf_copy_result_sampling_network(self.handle, _ptr(oracle), n_rays, _ptr(rgba8_out)))

    def render_numpy(self):
        """Convenience for tests/tools: renders and returns (rgb fp32 [R,3], rgba8 [R,4], Stats)."""
        n = self.info.rays_local
        if not hasattr(self, "_o_rgb") or self._o_rgb.shape[0] != n:
            self._o_rgb = self.empty((n, 3), np.float32)
            self._o_rgba = self.empty((n, 4), np.uint8)
        st = self.render(self._o_rgba, self._o_rgb, stats=True)
        return self._o_rgb.numpy(), self._o_rgba.numpy(), st

    def sync(self):
        self._check(self.lib.adanerf_sync(self.handle))

    def set_stream(self, hip_stream: Optional[int]):
        """Enqueue on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""
        self._check(self.lib.adanerf_set_stream(self.handle, hip_stream))

    def set_profiling(self, enabled: bool):
        self._check(self.lib.adanerf_set_profiling(self.handle, 1 if enabled else 0))

    def collect_stats(self):
        """(Stats summed over the frames rendered since the last collect, number of frames)."""
        st = Stats()
        n = C.c_int32(0)
        self._check(self.lib.adanerf_collect_stats(self.handle, C.byref(st), C.byref(n)))
        return st, n.value

    def assemble_strips(self, gathered, image_out):
        self._check(self.lib.adanerf_assemble_strips(self.handle, _ptr(gathered), _ptr(image_out)))

    def buffer(self, which: int, dtype, shape) -> np.ndarray:
        """Copies an internal buffer of the last rendered batch to the host."""
        p = C.c_void_p()
        nb = C.c_size_t()
        self._check(self.lib.adanerf_get_buffer(self.handle, which, C.byref(p), C.byref(nb)))
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= nb.value, (out.nbytes, nb.value)
        if out.nbytes:
            self._check(self.lib.adanerf_memcpy_d2h(self.handle, out.ctypes.data, p.value, out.nbytes))
        return out

    # -- stage-level entry points --------------------------------------------------------------------
    def ray_features(self, first_ray: int, n_rays: int, features_out=None, rays_out=None):
        self._check(self.lib.adanerf_ray_features(self.handle, first_ray, n_rays, _ptr(features_out), _ptr(rays_out)))

    def sample_mlp(self, first_ray: int, n_rays: int, oracle_out=None, rays_out=None):
        self._check(self.lib.adanerf_sample_mlp(self.handle, first_ray, n_rays, _ptr(oracle_out), _ptr(rays_out)))

    def compact(self, oracle, n_rays: int, n_max: int, thr: float, ray_offsets, ray_counts, sample_key, sample_w, total):
        self._check(self.lib.adanerf_compact(self.handle, _ptr(oracle), n_rays, n_max, thr, _ptr(ray_offsets),
                                             _ptr(ray_counts), _ptr(sample_key), _ptr(sample_w), _ptr(total)))

    def compact_guarded(self, oracle_approx, oracle_exact, n_rays: int, n_max: int, thr: float, eps: float, ray_offsets, ray_counts,
                        sample_key, sample_w, total, refined, eps_pair: float = 0.0, audit_period: int = 0, audit_phase: int = 0, monitor=None,
                        audit_fill_cap: int = 0, audit_cycle: int = 0):
        """monitor: optional device uint32[5] the caller zeroed (largest error bits, rows beyond a bound, largest pair error bits,
        audit mismatches, audited rows)."""
        self._check(self.lib.adanerf_compact_guarded(self.handle, _ptr(oracle_approx), _ptr(oracle_exact), n_rays, n_max, thr, eps, eps_pair,
                                                     audit_period, audit_phase, audit_fill_cap, audit_cycle, _ptr(ray_offsets), _ptr(ray_counts), _ptr(sample_key),
                                                     _ptr(sample_w), _ptr(total), _ptr(refined), _ptr(monitor)))

    def refresh_info(self) -> "Info":
        """Re-reads adanerf_info (guard_eps moves when the band is calibrated or widened after a violation)."""
        self._check(self.lib.adanerf_get_info(self.handle, C.byref(self.info)))
        return self.info

    def calibrate_guard(self, n_poses: int = 8, seed: int = 1, install: bool = False, pair: bool = False):
        """Largest |plain-fp16 - split-precision| raw output over n_poses x 4096 calibration rays (pair=True: also the largest
        error of a (kept - candidate) difference, as a tuple); install=True makes ADANERF_GUARD_CALIB_MARGIN x those the context's
        bounds and writes the model's calibration record."""
        d, dp = C.c_float(0), C.c_float(0)
        self._check(self.lib.adanerf_calibrate_guard(self.handle, n_poses, seed, 1 if install else 0, C.byref(d), C.byref(dp)))
        if install:
            self._check(self.lib.adanerf_get_info(self.handle, C.byref(self.info)))
        return (float(d.value), float(dp.value)) if pair else float(d.value)

    def guard_calibration_file(self) -> str:
        """Path of the calibration record of this context's (model, N, threshold)."""
        n = self.lib.adanerf_guard_calibration_file(self.handle, None, 0)
        if n < 0:
            self._check(n)
        buf = C.create_string_buffer(n)
        self.lib.adanerf_guard_calibration_file(self.handle, buf, n)
        return buf.value.decode()

    def probe_mfma(self, operands: str = "relu", f16: bool = False, target_ms: float = 100.0):
        """(TFLOP/s, MHz) this device sustains on register-only 16-bit MFMA loops with "zero" / "constant" / "random" / "relu" operands
        (include/adanerf_hip.h adanerf_probe_mfma)."""
        tf, mhz = C.c_float(0), C.c_float(0)
        self._check(self.lib.adanerf_probe_mfma(self.handle, {"zero": 0, "constant": 1, "random": 2, "relu": 3}[operands], 1 if f16 else 0,
                                                target_ms, C.byref(tf), C.byref(mhz)))
        return float(tf.value), float(mhz.value)

    def shade_features(self, rays, sample_key, n_samples: int, features_out):
        self._check(self.lib.adanerf_shade_features(self.handle, _ptr(rays), _ptr(sample_key), n_samples, _ptr(features_out)))

    def shade_mlp(self, rays, sample_key, total, max_samples: int, raw_out, precision: int = -1):
        self._check(self.lib.adanerf_shade_mlp(self.handle, _ptr(rays), _ptr(sample_key), _ptr(total), max_samples,
                                               precision, _ptr(raw_out)))

    def shade_mlp_z(self, rays, sample_key, sample_z, total, max_samples: int, raw_out, precision: int = -1):
        self._check(self.lib.adanerf_shade_mlp_z(self.handle, _ptr(rays), _ptr(sample_key), _ptr(sample_z), _ptr(total),
                                                 max_samples, precision, _ptr(raw_out)))

    def sample_pdf(self, oracle, n_rays: int, n: int, ray_offsets, ray_counts, sample_key, sample_w, sample_z, total):
        self._check(self.lib.adanerf_sample_pdf(self.handle, _ptr(oracle), n_rays, n, _ptr(ray_offsets), _ptr(ray_counts),
                                                _ptr(sample_key), _ptr(sample_w), _ptr(sample_z), _ptr(total)))

    # vanilla NeRF (SAMPLER_COARSE_FINE): uniform coarse samples, the coarse network, the fine sampler
    def sample_uniform(self, first_ray: int, n_rays: int, rays, ray_offsets, ray_counts, sample_key, total):
        self._check(self.lib.adanerf_sample_uniform(self.handle, first_ray, n_rays, _ptr(rays), _ptr(ray_offsets), _ptr(ray_counts),
                                                    _ptr(sample_key), _ptr(total)))

    def shade_mlp_coarse(self, rays, sample_key, total, max_samples: int, raw_out, precision: int = -1):
        self._check(self.lib.adanerf_shade_mlp_coarse(self.handle, _ptr(rays), _ptr(sample_key), _ptr(total), max_samples, precision, _ptr(raw_out)))

    def sample_from_coarse(self, raw_coarse, rays, n_rays: int, ray_offsets, ray_counts, sample_key, sample_z, total):
        self._check(self.lib.adanerf_sample_from_coarse(self.handle, _ptr(raw_coarse), _ptr(rays), n_rays, _ptr(ray_offsets), _ptr(ray_counts),
                                                        _ptr(sample_key), _ptr(sample_z), _ptr(total)))

    def composite_classic(self, raw, sample_z, rays, n_rays: int, n: int, rgb_out=None, rgba8_out=None):
        self._check(self.lib.adanerf_composite_classic(self.handle, _ptr(raw), _ptr(sample_z), _ptr(rays), n_rays, n,
                                                       _ptr(rgb_out), _ptr(rgba8_out)))

    def composite(self, raw, sample_w, ray_offsets, ray_counts, n_rays: int, rgb_out=None, rgba8_out=None):
        self._check(self.lib.adanerf_composite(self.handle, _ptr(raw), _ptr(sample_w), _ptr(ray_offsets), _ptr(ray_counts),
                                               n_rays, _ptr(rgb_out), _ptr(rgba8_out)))
