"""Image evaluation over a DONeRF-style dataset directory with the MI355X renderer -- the counterpart of the
reference's ``src/evaluate.py`` "images" evaluation (``generate_data`` :164-342: render every test view, MSE /
PSNR against the ground-truth PNG, mean samples per ray) for an exported model directory.

    python -m adanerf_amd.evaluate <model_dir> <dataset_dir> [--set test] [--out DIR] [--video out.y4m] [--precision bf16]

Dataset layout (src/datasets.py:146-213, 361-365, 480-542): ``dataset_info.json`` (``resolution``,
``camera_angle_x``, ``view_cell_center``, ``view_cell_size`` ...), ``transforms_<set>.json`` with
``frames[i].file_path`` ("./test/00000") and ``frames[i].transform_matrix`` (4x4 camera-to-world; pose =
[:3, 3], rotation = [:3, :3]), images ``<file_path>.png``.
"""
import argparse
import json
import math
import os
import sys
from typing import List, Optional

import numpy as np

from .png import read_png, write_png
from .renderer import NeuralRenderer, Settings


def load_dataset(dataset_dir: str, set_name: str = "test"):
    with open(os.path.join(dataset_dir, "dataset_info.json")) as f:
        info = json.load(f)
    with open(os.path.join(dataset_dir, "transforms_%s.json" % set_name)) as f:
        tr = json.load(f)
    w, h = int(info["resolution"][0]), int(info["resolution"][1])
    frames = []
    for fr in tr["frames"]:
        m = np.array(fr["transform_matrix"], dtype=np.float32)
        rel = fr["file_path"][2:] if fr["file_path"].startswith("./") else fr["file_path"]   # datasets.py:362
        frames.append(dict(pose=m[:3, 3].copy(), rot=m[:3, :3].copy(), image=os.path.join(dataset_dir, rel + ".png")))
    return dict(w=w, h=h, fov=float(info["camera_angle_x"]), info=info), frames


class Y4mWriter:
    """Uncompressed YUV4MPEG2 (4:4:4, BT.601 full range) -- the headless counterpart of the reference's
    ``imageio.mimwrite(... .mp4, fps=30)`` (src/evaluate.py:289-292); no codec library exists in this image, and any
    player / ffmpeg reads .y4m."""

    def __init__(self, path: str, w: int, h: int, fps: int = 30):
        self.f = open(path, "wb")
        self.f.write(("YUV4MPEG2 W%d H%d F%d:1 Ip A1:1 C444 XCOLORRANGE=FULL\n" % (w, h, fps)).encode())
        self.w, self.h = w, h

    def add(self, rgb8: np.ndarray):
        """rgb8: uint8 [h, w, 3]"""
        c = rgb8.astype(np.float32)
        y = 0.299 * c[..., 0] + 0.587 * c[..., 1] + 0.114 * c[..., 2]
        u = -0.168736 * c[..., 0] - 0.331264 * c[..., 1] + 0.5 * c[..., 2] + 128.0
        v = 0.5 * c[..., 0] - 0.418688 * c[..., 1] - 0.081312 * c[..., 2] + 128.0
        self.f.write(b"FRAME\n")
        for p in (y, u, v):
            self.f.write(np.clip(np.rint(p), 0, 255).astype(np.uint8).tobytes())

    def close(self):
        self.f.close()


def psnr_from_mse(mse: float) -> float:
    """src/evaluate.py:49-54: 10 log10(1 / mse), mse over all 3*h*w values."""
    return float("inf") if mse == 0 else 10.0 * math.log10(1.0 / mse)


def evaluate(model_dir: str, dataset_dir: str, set_name: str = "test", out_dir: Optional[str] = None,
             precision: str = "bf16", batch_size: int = -1, max_frames: int = 0, quiet: bool = False,
             video: Optional[str] = None, fps: int = 30):
    meta, frames = load_dataset(dataset_dir, set_name)
    if max_frames > 0:
        frames = frames[:max_frames]
    w, h = meta["w"], meta["h"]
    results: List[dict] = []
    with NeuralRenderer(Settings(model_dir, w, h, batch_size=batch_size), precision=precision) as r:
        if abs(r.info.fov - meta["fov"]) > 1e-4 and not quiet:
            print("warning: dataset camera_angle_x %.6f differs from the model's fov %.6f (the model's is used)" %
                  (meta["fov"], r.info.fov), file=sys.stderr)
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
        vid = Y4mWriter(video, w, h, fps) if video else None
        for i, fr in enumerate(frames):
            r.set_camera(fr["pose"], fr["rot"])
            rgb, rgba, st = r.render_numpy()
            rec = dict(frame=i, image=fr["image"], samples_per_ray=st.total_samples / float(w * h), ms=st.ms_total)
            if os.path.exists(fr["image"]):
                gt = read_png(fr["image"])
                if gt.shape[0] != h or gt.shape[1] != w:
                    raise ValueError("%s: expected %dx%d, got %dx%d" % (fr["image"], w, h, gt.shape[1], gt.shape[0]))
                ref = gt[:, :, :3].astype(np.float32).reshape(-1, 3) / 255.0        # datasets.py:286-287
                mse = float(np.mean((rgb.astype(np.float64) - ref) ** 2))
                rec.update(mse=mse, psnr=psnr_from_mse(mse))
            if out_dir:
                write_png(os.path.join(out_dir, "%05d.png" % i), rgba[:, :3].reshape(h, w, 3))
            if vid:
                vid.add(rgba[:, :3].reshape(h, w, 3))
            results.append(rec)
            if not quiet:
                print("frame %d: %s" % (i, ", ".join("%s=%s" % (k, ("%.4f" % v) if isinstance(v, float) else v)
                                                      for k, v in rec.items() if k not in ("frame", "image"))))
        if vid:
            vid.close()
    with_gt = [x for x in results if "psnr" in x]
    summary = dict(frames=len(results), mean_samples_per_ray=float(np.mean([x["samples_per_ray"] for x in results])) if results else 0.0,
                   mean_ms=float(np.mean([x["ms"] for x in results])) if results else 0.0)
    if with_gt:
        summary.update(mean_psnr=float(np.mean([x["psnr"] for x in with_gt])), mean_mse=float(np.mean([x["mse"] for x in with_gt])))
    return summary, results


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("model_dir")
    ap.add_argument("dataset_dir")
    ap.add_argument("--set", default="test")
    ap.add_argument("--out", default=None, help="write the rendered frames as PNG here")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--batch-size", type=int, default=-1)
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--video", default=None, help="also write the rendered frames as an uncompressed .y4m video")
    ap.add_argument("--fps", type=int, default=30)
    a = ap.parse_args(argv)
    summary, _ = evaluate(a.model_dir, a.dataset_dir, a.set, a.out, a.precision, a.batch_size, a.max_frames, video=a.video, fps=a.fps)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
