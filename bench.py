#!/usr/bin/env python3
"""bench.py -- AdaNeRF frame throughput on MI355X (BASELINE.json metric: FPS at 800x800).

One "step" = one full frame of the hot path (ray generation -> sampling MLP -> adaptive compaction ->
fused PE + shading MLP -> compositing [-> RCCL gather of the RGBA8 strips on N > 1]) for a fixed camera
on synthetic 800x800 ray batches.  Workload at every N: BASELINE.json configs[1]
(800x800, N = 8, threshold 0.2, bf16 shading MLP; weights = the reference's exported
`sample_pavillon_16` model, carried as a data fixture, random-init fallback if absent).

  python bench.py --gpus 1 --steps 30 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see README / DESIGN.md for the field meanings).
"""
import argparse
import json
import re
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required for RCCL between processes on these hosts

SHADE_FLOP_PER_SAMPLE = 1186816      # 2 x 593408 MAC
SAMPLE_FLOP_PER_RAY = 898048         # 2 x 449024 MAC
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # MI355X dense MFMA peaks (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (width, height, N, threshold, weights tag)
    "config2": (800, 800, 8, 0.2, "sample_pavillon_16"),
    "config3_dense": (800, 800, 128, 0.0, "sample_pavillon_16"),
    "config4": (800, 800, 8, 0.1, "sample_pavillon_16"),
    # BASELINE configs[4]: 1920x1080 LLFF-NDC scene (2-2 oracle encoding, linear depth, no normalisation); no NDC
    # weights ship with the reference -> seeded random-init weights; sweep the threshold with --threshold
    "config5_ndc": (1920, 1080, 8, 0.2, "ndc_random_init"),
    # vanilla NeRF with hierarchical sampling, the original paper's 64 coarse + 128 more samples per ray through two
    # 8 x 256 networks (64 + 192 = 256 network evaluations per ray); random-init weights.  Not a BASELINE configuration.
    "nerf_coarse_fine": (800, 800, 128, 1.0, "nerf_pair_random_init"),
    # a topology other than 8 x 256 / skip 4 -- both networks 6 x 128 (skip 2), random init.  The sampling network runs
    # on the run-time-shaped exact-fp32 kernel, the shading network on the run-time-shaped 16-bit kernel.  Not a BASELINE configuration.
    "generic_6x128": (800, 800, 8, 0.65, "generic_6x128_random_init"),
    "generic_4x64": (800, 800, 8, 0.65, "generic_4x64_random_init"),        # the other two widths of the run-time-shaped kernels
    "generic_5x256": (800, 800, 8, 0.65, "generic_5x256_random_init"),
}


def generic_shape(tag):
    """(layers, width, skip) of a generic_<L>x<W>_random_init workload, None for the others"""
    m = re.match(r"generic_(\d+)x(\d+)_random_init$", tag)
    return (int(m.group(1)), int(m.group(2)), int(m.group(1)) // 3) if m else None


def shade_flop_per_sample(tag):
    """2 x MACs of the shading network per sample (SURVEY 8d counts the 8 x 256 one: 593 408 MAC)"""
    if generic_shape(tag):
        d, w, skip = generic_shape(tag)
        n_pos, n_dir = 63, 27
        mac = n_pos * w + sum((w + n_pos if i == skip + 1 else w) * w for i in range(1, d)) + w * w + w + (w + n_dir) * (w // 2) + (w // 2) * 3
        return 2 * mac
    return SHADE_FLOP_PER_SAMPLE


def build_model_dir(td, tag, n, thr):
    """Model directory for the GPU path, written with the package's own format tools (adanerf_amd.modeldir)."""
    from adanerf_amd import modeldir as M
    gold = os.path.join(ROOT, "tests", "golden")
    wpath = os.path.join(gold, "weights_%s.npz" % tag)
    spath = os.path.join(gold, "scene_%s.json" % tag)
    if os.path.exists(wpath) and os.path.exists(spath):
        z = np.load(wpath)
        n0 = {k[3:]: z[k] for k in z.files if k.startswith("n0/")}
        n1 = {k[3:]: z[k] for k in z.files if k.startswith("n1/")}
        s = json.load(open(spath))
        data = "synthetic rays; weights = reference's exported %s model (fixture)" % tag
    elif tag == "ndc_random_init":
        n0, n1 = M.random_init_weights(7, n_in0=30, oracle_bias=-0.55, oracle_scale=0.5)
        s = dict(view_cell_center=(0.0, 0.0, 0.0), view_cell_size=(2.0, 2.0, 1.0), depth_range=(0.9, 12.0), fov=1.0, max_depth=12.0)
        data = "synthetic rays; random-init weights (seed 7), NDC / linear depth / 2-2 oracle encoding"
    elif generic_shape(tag):
        gl, gw, gs = generic_shape(tag)
        n0, n1 = M.random_init_weights(21, layers=(gl, gl), widths=(gw, gw), skip1=gs)
        s = dict(view_cell_center=(0.783, -3.19, 1.39), view_cell_size=(0.7, 0.7, 0.2),
                 depth_range=(0.1542200982570648, 8.358194804191589), fov=1.1386263370513916, max_depth=8.79825210571289)
        data = "synthetic rays; random-init weights (seed 21), both networks %d x %d / skip %d" % (gl, gw, gs)
    elif tag == "nerf_pair_random_init":
        n0, n1 = M.random_init_nerf_pair(3)
        s = dict(view_cell_center=(0.783, -3.19, 1.39), view_cell_size=(0.7, 0.7, 0.2),
                 depth_range=(0.1542200982570648, 8.358194804191589), fov=1.1386263370513916, max_depth=8.79825210571289)
        data = "synthetic rays; two random-init NeRF nets (seed 3): vanilla NeRF, 64 coarse + 128 fine samples per ray"
    else:
        n0, n1 = M.random_init_weights(0)
        s = dict(view_cell_center=(0.783, -3.19, 1.39), view_cell_size=(0.7, 0.7, 0.2),
                 depth_range=(0.1542200982570648, 8.358194804191589), fov=1.1386263370513916, max_depth=8.79825210571289)
        data = "synthetic rays; random-init weights (seed 0)"
    scene = dict(view_cell_center=s["view_cell_center"], view_cell_size=s["view_cell_size"], depth_range=s["depth_range"],
                 fov=s["fov"], max_depth=s["max_depth"], num_samples=n, threshold=thr)
    if tag == "ndc_random_init":
        scene.update(use_ndc=True, depth_transform="linear", pos_enc=((2, 2), (10, 4)), normalization="None")
    if tag == "nerf_pair_random_init":
        scene.update(num_samples_coarse=64, accumulation_mult="")
    M.write_model_dir(td, scene, n0, n1)
    return scene, data


def cpu_baseline(model_dir, w, h, pose, rot, budget_s=10.0):
    """The oracle (numpy port of the reference's PyTorch path with torch's CPU GEMM, validated against the golden vectors)
    timed on this box's host cores: P worker processes x T threads, each rendering its own band of rows of the same frame
    AT THE SAME TIME (oracle/cpu_worker.py), i.e. one parallel CPU render of (a bounded part of) the frame.  The reference's
    own CPU path is one PyTorch process; on many-core hosts a single process leaves most cores idle in the element-wise
    stages (encodings, sort), so the port is run the way a CPU deployment would run it.  `cores` = P x T threads actually
    computing.  This leg (and the quality check that reuses its output) is the ONLY place bench.py touches oracle/."""
    import shutil
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import adanerf_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    # Threads in total = half the logical CPUs (the hosts of this pool are SMT-2: 256 logical = 128 cores; measured on one,
    # profiles/r03_cpu_baseline_split.log: 64 x 2 threads 0.119 frames/s, 32 x 4 0.113, 64 x 4 0.088, 128 x 2 0.081, 16 x 16 0.058)
    T = 2 if avail >= 8 else 1
    P = max(1, min((avail // 2) // T if avail >= 8 else avail, 64, h // 2))
    if os.environ.get("ADANERF_CPU_PT"):      # "P,T": try another split of the host's cores (tools/r03_call11.sh)
        P, T = (int(x) for x in os.environ["ADANERF_CPU_PT"].split(","))
    band = h // P
    sync = tempfile.mkdtemp(prefix="adanerf_cpu_")
    cam = os.path.join(sync, "cam.npy")
    np.save(cam, np.concatenate([np.asarray(pose, np.float32).reshape(3), np.asarray(rot, np.float32).reshape(9)]))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "oracle"), os.environ.get("PYTHONPATH", "")]))
    procs = []
    for i in range(P):
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), model_dir, str(w), str(h),
                                       str(i * band), str(band), str(T), str(budget_s), sync, str(i), cam], env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
    t_wait = time.time()
    while sum(os.path.exists(os.path.join(sync, "ready.%d" % i)) for i in range(P)) < P:
        if any(p.poll() not in (None, 0) for p in procs) or time.time() - t_wait > 600:
            err = b"".join(p.stderr.read() for p in procs if p.poll() not in (None, 0))
            for p in procs:
                p.kill()
            raise RuntimeError("cpu_baseline worker failed: %s" % err[-2000:].decode(errors="replace"))
        time.sleep(0.05)
    open(os.path.join(sync, "go"), "w").close()
    for p in procs:
        p.wait()
    outs = [np.load(os.path.join(sync, "out.%d.npz" % i)) for i in range(P)]
    rows_done = int(sum(int(o["rows_done"]) for o in outs))
    wall = max(float(o["seconds"]) for o in outs)
    fps = (rows_done / float(h)) / wall
    # what the quality check compares against the GPU frame: EVERY row the workers finished (round 6; one band of 12 rows = 9 600 rays before).  `sel` =
    # indices of those rays in the frame, in band order
    done = [(int(o["row0"]), int(o["rows_done"])) for o in outs]
    res = {"rgb": np.concatenate([o["rgb"][:n * w] for o, (_, n) in zip(outs, done)]),
           "count": np.concatenate([o["count"][:n * w] for o, (_, n) in zip(outs, done)]),
           "sel": np.concatenate([np.arange(r0 * w, (r0 + n) * w) for r0, n in done])}
    row0, rows = done[P // 2]
    spp = float(np.mean(np.concatenate([o["count"] for o in outs])))
    backend = str(outs[0]["backend"])
    shutil.rmtree(sync, ignore_errors=True)
    return {"value": fps, "unit": "frames/s", "cores": int(P * T), "kind": "port",
            "sample": "%d of %d image rows (%d rays, %.2f samples/ray) of the same frame in %.1f s: %d processes x %d threads, "
                      "numpy + %s fp32, %d host cores available" % (rows_done, h, rows_done * w, spp, wall, P, T, backend, avail)}, \
        res, (row0, rows), O.psnr


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same
    arguments>` (one rank per GPU, rendezvous on 127.0.0.1 at a free port).  Does not return."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a launcher -> %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


class Watchdog:
    """A rank that is stuck in a collective never prints.  Every phase has `seconds` to finish (the timer is re-armed whenever `phase` is set: a cold
    build or a long extra measurement does not use up the exchange's time); when a phase runs out every rank leaves through os._exit and rank 0
    first prints the line it has: the finished record if the main measurement is done (a hang in one of the extra measurements behind it), else a
    line with `value` null and the reason -- the render-only rate of phase A rides along as `render_only` (it is NOT an N-GPU frames/s: no payload
    was gathered, no frame assembled, so it never appears as `value`) -- and the process exits with status 3."""

    def __init__(self, seconds, rank):
        self.rank, self.seconds, self.record, self.fallback, self._phase, self.timer = rank, seconds, None, None, "start", None
        self._arm()

    def _arm(self):
        import threading
        if self.timer:
            self.timer.cancel()
        self.timer = threading.Timer(self.seconds, self.fire) if self.seconds > 0 else None
        if self.timer:
            self.timer.daemon = True
            self.timer.start()

    @property
    def phase(self):
        return self._phase

    @phase.setter
    def phase(self, name):
        self._phase = name
        self._arm()

    def fire(self):
        rec = self.record or self.fallback
        if self.rank == 0:
            why = "watchdog: phase '%s' made no progress for %d s" % (self._phase, self.seconds)
            if rec is None:
                rec = {"metric": "FPS at 800x800", "value": None, "unit": "frames/s", "error": why}
            elif rec is self.fallback:
                rec["error"] = rec["config"]["exchange"]["error"] = why
            else:
                rec.setdefault("notes", []).append(why)
            sys.stdout.write(json.dumps(rec) + "\n")
            sys.stdout.flush()
        # (a rank other than 0 leaves quietly: the launcher then still waits for rank 0 and returns ITS status)
        os._exit(0 if self.rank != 0 or rec.get("value") is not None else 3)

    def cancel(self):
        if self.timer:
            self.timer.cancel()


class GpuTelemetry:
    """Shader clock and socket power of one GPU, sampled in a thread around the timed region through librocm_smi64 (sysfs is not visible in the
    containers of this pool; rocm-smi's library is): lets a reader of the line tell a slow box from a slow kernel next to roofline.sustained_peak.
    Everything is optional: what cannot be read is reported as null with the reason."""

    def __init__(self, device_index=0, period=0.02):
        import ctypes as C
        import threading
        self.C, self.dev, self.period, self.samples, self.err, self.lib = C, device_index, period, [], None, None
        self._stop, self._thread = threading.Event(), None
        try:
            self.lib = C.CDLL("librocm_smi64.so")
            rc = self.lib.rsmi_init(C.c_uint64(0))
            if rc != 0:
                self.err, self.lib = "rsmi_init: status %d" % rc, None
        except OSError as e:
            self.err = "librocm_smi64.so: %s" % e

    class _Freqs(__import__("ctypes").Structure):
        _fields_ = [("has_deep_sleep", __import__("ctypes").c_bool), ("num_supported", __import__("ctypes").c_uint32),
                    ("current", __import__("ctypes").c_uint32), ("frequency", __import__("ctypes").c_uint64 * 33)]

    def read(self):
        C = self.C
        mhz = watts = None
        f = self._Freqs()
        if self.lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(self.dev), C.c_int(0), C.byref(f)) == 0 and f.current < 33:      # RSMI_CLK_TYPE_SYS
            mhz = f.frequency[f.current] / 1e6
        p = C.c_uint64(0)
        if self.lib.rsmi_dev_current_socket_power_get(C.c_uint32(self.dev), C.byref(p)) == 0:
            watts = p.value / 1e6
        return mhz, watts

    def start(self):
        import threading
        if not self.lib:
            return self

        def loop():
            while not self._stop.is_set():
                try:
                    self.samples.append(self.read())
                except Exception as e:      # noqa: BLE001 -- telemetry must never take the measurement with it
                    self.err = "rsmi read: %s" % e
                    return
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread:
            self._stop.set()
            self._thread.join(timeout=2.0)      # a management-library call that hangs must not hang the bench: the daemon thread is abandoned
            if self._thread.is_alive():
                return {"sclk_mhz_mean": None, "power_w_mean": None, "error": "librocm_smi64 call did not return", "samples": len(self.samples)}
        if not self.lib:
            return {"sclk_mhz_mean": None, "power_w_mean": None, "error": self.err}
        clk = [m for m, _ in self.samples if m]
        pw = [w for _, w in self.samples if w]
        bdf = self.C.c_uint64(0)
        self.lib.rsmi_dev_pci_id_get(self.C.c_uint32(self.dev), self.C.byref(bdf))
        return {"sclk_mhz_mean": sum(clk) / len(clk) if clk else None, "sclk_mhz_min": min(clk) if clk else None, "sclk_mhz_max": max(clk) if clk else None,
                "power_w_mean": sum(pw) / len(pw) if pw else None, "power_w_max": max(pw) if pw else None, "samples": len(self.samples),
                "source": "librocm_smi64 (rsmi_dev_gpu_clk_freq_get SYS, rsmi_dev_current_socket_power_get), device %d, every %d ms over the timed steps"
                          % (self.dev, int(self.period * 1e3)), "pci_bdf": "%04x:%02x:%02x.%x" % ((bdf.value >> 32) & 0xffff, (bdf.value >> 8) & 0xff, (bdf.value >> 3) & 0x1f, bdf.value & 7)}


def run_peer(args):
    """--exchange peer: ONE process owns all N GPUs (what `adanerf --gpus N` does, adanerf_amd/host/neuralrenderer.cpp): a context per
    (GPU, sub-share), every context renders its strips on its own stream, each payload is copied to GPU 0 behind its render
    (adanerf_gather_to = hipMemcpyPeerAsync over xGMI, ordered by events), GPU 0 de-interleaves; one host sync per frame.  No RCCL, no
    torch.  Prints one JSON line of the same shape; `value` = frames / s of this path."""
    import adanerf_amd
    from adanerf_amd import build as B
    from adanerf_amd import modeldir as M
    from adanerf_amd import sharding
    B.build_library()
    n = args.gpus
    one_dev = os.environ.get("ADANERF_BENCH_ONE_DEVICE") == "1"
    w, h, n_max, thr, tag = WORKLOADS[args.workload]
    if args.threshold is not None:
        thr = args.threshold
    td = tempfile.mkdtemp(prefix="adanerf_bench_peer_")
    scene, data = build_model_dir(td, tag, n_max, thr)
    pose = np.array(scene["view_cell_center"], dtype=np.float32)
    rot = M.camera_rotation(100.0, 0.0) if tag != "ndc_random_init" else np.eye(3, dtype=np.float32)
    P = args.sub_shares
    if not P:
        even = any(h % sr == 0 and (h // sr) % (2 * n) == 0 for sr in range(1, 9))
        P = 2 if n > 1 and even else 1
    vworld = n * P
    strip_rows = sharding.balanced_strip_rows(h, vworld)
    rs = []
    for v in range(vworld):
        q = adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h, batch_size=args.batch_rays), precision=args.precision, sampling=args.sampling,
                                       guard_eps=args.guard_eps, guard_audit_period=args.guard_audit_period, device_id=0 if one_dev else v // P,
                                       shard_rank=v, shard_world=vworld, strip_rows=strip_rows)
        q.init()
        q.set_camera(pose, rot)
        rs.append(q)
    root = rs[0]
    stride = root.info.rays_local_max * 4
    gathered = root.empty((vworld, root.info.rays_local_max, 4), np.uint8)
    image = root.empty((w * h, 4), np.uint8)
    payloads = [None] + [q.empty((q.info.rays_local_max, 4), np.uint8) for q in rs[1:]]

    def step():
        root.render(gathered.ptr, None)                      # virtual rank 0 renders straight into its slot
        for v in range(1, vworld):
            rs[v].render(payloads[v], None)
            root.gather_from(gathered.ptr + v * stride, rs[v], payloads[v], stride)
        root.assemble_strips(gathered, image)
        root.sync()                                          # the frame is complete on GPU 0

    for _ in range(args.warmup):
        step()
    for q in rs:
        q.sync()
    for q in rs:
        q.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for q in rs:
        q.sync()
    dt = time.perf_counter() - t0
    samples, shade_ms, per_dev = 0.0, [], [0.0] * n
    for v, q in enumerate(rs):
        st, frames = q.collect_stats()
        q.set_profiling(False)
        samples += st.total_samples / max(frames, 1)
        shade_ms.append(st.ms_shade_mlp / max(frames, 1))
        per_dev[v // P] += st.total_samples / max(frames, 1)
    if args.dump_image:
        np.save(args.dump_image, image.numpy().reshape(h, w, 4))
    mean_s = sum(per_dev) / len(per_dev)
    rec = {"metric": "FPS at %dx%d" % (w, h), "value": args.steps / dt, "unit": "frames/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.precision,
           "data": data,
           "config": {"workload": "%s: %dx%d, N=%d, threshold %.2f, 8x256 shading MLP %s, sampling MLP %s" % (args.workload, w, h, n_max, thr, args.precision, args.sampling),
                      "parallelism": "image-strip shard x%d (%d-row strips, round-robin over %d virtual ranks, %d concurrent sub-share(s) per GPU), ONE host "
                                     "process, payloads copied to GPU 0 with hipMemcpyPeerAsync (adanerf_gather_to), one host sync per frame" % (n, strip_rows, vworld, P),
                      "exchange": {"mode": "peer", "backend": "hipMemcpyPeerAsync", "rccl_ranks": 0, "world_size": n, "payload_bytes_per_rank": int(stride * P),
                                   "sub_shares_per_rank": P, "all_contexts_on_device_0": one_dev},
                      "sub_shares_per_gpu": P, "samples_per_frame": samples, "mean_samples_per_ray": samples / (w * h)},
           "shards": {"samples_per_frame": per_dev, "shade_ms_per_frame_per_context": shade_ms,
                      "sample_imbalance_max_over_mean": max(per_dev) / mean_s if mean_s > 0 else None}}
    print(json.dumps(rec))
    sys.stdout.flush()
    for q in rs:
        q.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--batch-rays", type=int, default=-1)
    ap.add_argument("--threshold", type=float, default=None, help="override the workload's adaptive sampling threshold")
    ap.add_argument("--sampling", default=None, choices=["split", "fp32", "fp16", "guarded"],
                    help="sampling-MLP arithmetic: split-fp16 (default: fp32-accurate on every ray, selections exact by construction), exact fp32, "
                         "guarded (opt-in: plain fp16 + split-fp16 on the rays inside the audited guard band -- the split engine's selections while the "
                         "measured band holds; reported beside the headline as guarded_mode), or the opt-in plain fp16 speed mode (speed_mode)")
    ap.add_argument("--guard-eps", type=float, default=0.0, help="band of --sampling guarded (0: the model's calibration record / measured at the first frame)")
    ap.add_argument("--guard-audit-period", type=int, default=0, help="--sampling guarded: audit 1 / period of all rays per frame (0: library default 16, < 0: off)")
    ap.add_argument("--dry-run", action="store_true", help="N > 1: initialise the process groups (gloo control plane, RCCL data plane), do ONE gather of the real payload "
                                                           "size, print what every rank saw (device, PCI bus id, memory, the gather's time) as one JSON line and stop before any render")
    ap.add_argument("--no-sustained-probe", action="store_true", help="skip the ~0.5 s MFMA-rate probe behind roofline.sustained_peak / frac_of_sustained")
    ap.add_argument("--no-exact-mode", action="store_true", help="--sampling guarded: skip the extra every-ray-split-precision measurement reported under exact_mode")
    ap.add_argument("--no-guarded-mode", action="store_true", help="skip the extra guarded-sampling measurement reported under guarded_mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-speed-mode", action="store_true", help="skip the extra fp16-sampling measurement reported under speed_mode")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of wall time the CPU baseline may compute (all its cores busy)")
    ap.add_argument("--orbit", type=int, default=0,
                    help="K > 0: step i renders pose i %% K of a K-pose orbit inside the view cell (yaw and position vary) instead of "
                         "the fixed camera; quality / cpu_baseline still refer to pose 0")
    ap.add_argument("--sub-shares", type=int, default=0, choices=[0, 1, 2],
                    help="N > 1: every rank renders its share of a frame as this many sub-shares on as many contexts / streams, concurrently "
                         "(0: 2 for N > 1, else 1); the same frame, so a frame's latency does not grow")
    ap.add_argument("--no-split-mode", action="store_true", help="N = 1: skip the extra two-sub-shares measurement reported under split_frame_mode")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=[0, 1, 2],
                    help="N > 1 only: 1 (default) renders one frame at a time, as on a single GPU and as an interactive viewer needs it; "
                         "2 renders alternate frames on two contexts / streams of the rank, so that the tail rounds, ring prologues and "
                         "launch gaps of one frame's kernels overlap the next frame's (more throughput; a frame's latency doubles)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "all_gather", "gloo", "peer"],
                    help="N > 1, how the RGBA8 strip payloads reach GPU 0: rccl (default) = one torch.distributed.gather on the RCCL group per frame "
                         "(falls back to all_gather, then to a host-staged gloo gather, if RCCL refuses -- the line says which ran); all_gather = "
                         "all_gather_into_tensor on the RCCL group; gloo = host-staged; peer = ONE process owning all GPUs, hipMemcpyPeerAsync to GPU 0 "
                         "(adanerf_gather_to, the C++ host's path) -- no launcher needed")
    ap.add_argument("--no-alternatives", action="store_true",
                    help="N > 1: skip the short extra measurements of the other exchange paths reported under config.exchange.alternatives")
    ap.add_argument("--watchdog", type=float, default=float(os.environ.get("ADANERF_BENCH_WATCHDOG_S", "600")),
                    help="seconds after which a run that is stuck (a collective that never returns) prints the line it has and exits; 0: off")
    ap.add_argument("--dump-image", default=None, help="rank 0 writes the last frame's RGBA8 image [h,w,4] as .npy (tests)")
    args = ap.parse_args()
    if args.sampling is None:
        args.sampling = "split"      # DESIGN 1, the default rule: exact by construction unless guarded is >= 8 % faster in this very run (4-5 %)

    launched = "RANK" in os.environ                      # under torch.distributed.run (the driver's N > 1 line)
    if args.exchange == "peer":
        # one process, all GPUs.  Under a launcher rank 0 is that process and the other ranks have nothing to do.
        if not launched or int(os.environ.get("RANK", "0")) == 0:
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
                os.environ.pop(k, None)
            run_peer(args)
        return
    if args.gpus > 1 and not launched:
        relaunch_under_torchrun(args.gpus)                # plain `python bench.py --gpus N` launches itself

    import torch
    import adanerf_amd
    from adanerf_amd import build as B
    from adanerf_amd.renderer import GUARD_FROM as R_GUARD_FROM
    from adanerf_amd import modeldir as M

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    dog = Watchdog(args.watchdog if world > 1 or launched else 0, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # Validation hooks for a 1-GPU box (tests/test_gpu_parity.py::test_bench_two_ranks_share_one_gpu): all ranks on
    # device 0 and gloo for the exchange, because RCCL refuses two ranks on one device.  Never set by the driver.
    backend = os.environ.get("ADANERF_BENCH_DIST_BACKEND", "nccl")
    if args.exchange == "gloo":
        backend = "gloo"
    if os.environ.get("ADANERF_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # ADANERF_BENCH_FORCE_DIST=1 (tests): take the process-group / gather / assemble path even at world size 1, so the
    # RCCL branch runs on a 1-GPU box exactly as it does on N > 1 (launch under torch.distributed.run --nproc-per-node 1)
    use_dist = world > 1 or (os.environ.get("ADANERF_BENCH_FORCE_DIST") == "1" and launched)
    # Two groups: the CONTROL plane (barriers, max-over-ranks timing, per-rank statistics: a few scalars on the host) is always gloo, the DATA
    # plane (one exchange of the RGBA8 strip payloads per frame) is RCCL.  A data plane that fails -- RCCL with more than one rank has never
    # run on the builder's side -- then cannot take the line with it: the ranks agree over gloo on what failed and fall back.
    data_pg, xerrors = None, []
    if use_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dog.phase = "process group init (gloo)"
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=max(300.0, 2.0 * args.watchdog)))
        if backend == "nccl":
            dog.phase = "process group init (RCCL)"
            try:
                data_pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180), device_id=torch.device("cuda", local_rank))
            except Exception as e:      # noqa: BLE001 -- anything RCCL raises here is reported, not fatal
                xerrors.append("new_group(nccl): %s: %s" % (type(e).__name__, str(e)[:300]))
            # a communicator that came up on some ranks only is of no use to anybody: every rank keeps it, or none does
            have = torch.tensor([1 if data_pg is not None else 0], dtype=torch.int32)
            dist.all_reduce(have, op=dist.ReduceOp.MIN)
            if not bool(have.item()) and data_pg is not None:
                xerrors.append("new_group(nccl): failed on another rank")
                data_pg = None
    if rank == 0:
        B.build_library()
    if dist:
        dist.barrier()
    if args.dry_run:
        # What does the node look like to this job, before any render?  Every rank reports its device; the data plane moves one payload of the size a
        # frame's share has (RGBA8 strips, padded to the largest share) three times.  A hang ends in the watchdog's line (value null, status 3).
        import socket
        from adanerf_amd import sharding as _sh
        w_, h_ = WORKLOADS[args.workload][0], WORKLOADS[args.workload][1]
        sr_ = _sh.balanced_strip_rows(h_, world)
        payload_bytes = -(-(h_ // sr_) // world) * sr_ * w_ * 4
        prop = torch.cuda.get_device_properties(local_rank)
        free_b, total_b = torch.cuda.mem_get_info(local_rank)
        me = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "pid": os.getpid(), "device": prop.name, "arch": getattr(prop, "gcnArchName", None),
              "compute_units": prop.multi_processor_count, "pci_bus_id": "%04x:%02x:%02x" % (getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_bus_id", 0), getattr(prop, "pci_device_id", 0)),
              "memory_free_GiB": free_b / 2 ** 30, "memory_total_GiB": total_b / 2 ** 30}
        ranks_seen = [me]
        res = {"backend": backend if data_pg is not None else "gloo", "payload_bytes_per_rank": payload_bytes, "ms": [], "error": None, "payloads_intact": None, "errors": xerrors}
        if dist:
            dog.phase = "dry run: rank reports (gloo)"
            ranks_seen = [None] * world
            dist.all_gather_object(ranks_seen, me)
            dog.phase = "dry run: one gather of the real payload size (%s)" % res["backend"]
            try:
                dev = "cuda" if data_pg is not None else "cpu"
                mine = torch.full((payload_bytes,), rank % 251, dtype=torch.uint8, device=dev)
                recv = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    dist.gather(mine, recv, dst=0, group=data_pg)
                    torch.cuda.synchronize()
                    res["ms"].append((time.perf_counter() - t0) * 1e3)
                if rank == 0:
                    res["payloads_intact"] = all(int(recv[k][0]) == k % 251 and int(recv[k][-1]) == k % 251 for k in range(world))
            except Exception as e:      # noqa: BLE001
                res["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
            dog.phase = "dry run: closing barrier"
            dist.barrier()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "workload": args.workload, "ranks": ranks_seen, "exchange": res,
                              "distinct_devices": len({(x["host"], x["pci_bus_id"]) for x in ranks_seen}) if all(ranks_seen) else None}))
        dog.cancel()
        return

    def all_ok(ok):
        """every rank's flag, agreed over the control plane"""
        if not dist:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    w, h, n_max, thr, tag = WORKLOADS[args.workload]
    generic_wl = args.workload.startswith("generic_")
    if args.threshold is not None:
        thr = args.threshold
    td = tempfile.mkdtemp(prefix="adanerf_bench_%d_" % rank)
    scene, data = build_model_dir(td, tag, n_max, thr)
    pose = np.array(scene["view_cell_center"], dtype=np.float32)
    rot = M.camera_rotation(100.0, 0.0) if tag != "ndc_random_init" else np.eye(3, dtype=np.float32)   # LLFF: looking down -z

    from adanerf_amd import sharding
    # N > 1: what does not shrink with N (a kernel's last, partly filled round, the weight ring's prologue, dependent-launch gaps) is
    # ~15 % of an 80 000-ray share.  Each rank therefore renders its share of a frame as P sub-shares -- virtual ranks
    # rank * P .. rank * P + P - 1 of a world of N * P -- on P contexts / streams at once: one sub-share's kernels take the CUs the
    # other's tails leave idle.  It is the SAME frame on every stream, so a frame's latency does not grow (measured: it shrinks,
    # tools/probes/split_shares.py); the gathered payloads are already in virtual-rank order.
    fif_req = args.frames_in_flight or int(os.environ.get("ADANERF_BENCH_FRAMES_IN_FLIGHT", "0")) or 1
    P = args.sub_shares or int(os.environ.get("ADANERF_BENCH_SUB_SHARES", "0"))
    if not P:
        # default: two sub-shares per rank where the image rows split evenly over 2 N virtual ranks (800 rows do for N = 2, 4, 8; 1080 rows
        # over 16 do not: 6 % more rays on the largest sub-share cost more than the overlap returns, profiles/r04_split_shares_*.log)
        even = any(h % sr == 0 and (h // sr) % (2 * world) == 0 for sr in range(1, 9))
        P = 2 if world > 1 and fif_req == 1 and even else 1
    if not use_dist:
        P = 1
    vworld = world * P
    strip_rows = sharding.balanced_strip_rows(h, vworld)

    def new_renderer(vrank):
        q = adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h, batch_size=args.batch_rays), precision=args.precision, sampling=args.sampling,
                                       guard_eps=args.guard_eps, guard_audit_period=args.guard_audit_period, device_id=local_rank, shard_rank=vrank,
                                       shard_world=vworld, strip_rows=strip_rows)
        q.init()
        q.set_camera(pose, rot)
        return q

    r = new_renderer(rank * P)
    # N > 1: a share of the frame is a few rounds of each kernel's persistent grid, and what does not shrink with N (the last,
    # partly filled round, the weight ring's prologue, dependent-launch gaps) is ~15 % of it at N = 8.  Two frames in
    # flight on two contexts / streams let the next frame's kernels take the CUs a kernel's tail leaves idle.
    fif = fif_req
    if not use_dist or P > 1:
        fif = 1                                   # sub-shares and a second frame in flight are alternatives
    rs = [r]
    if fif == 2:
        rs.append(new_renderer(rank))
    for k in range(1, P):
        rs.append(new_renderer(rank * P + k))
    dev = torch.device("cuda", local_rank)
    # Streams: the renderer enqueues on `tstream`; on N > 1 the exchange of frame k (RGBA8 strip payloads -> rank 0
    # over RCCL/xGMI) runs on `cstream` behind an event, so it overlaps the render of frame k+1.  Payload and gather
    # buffers are double-buffered; rank 0 de-interleaves frame k (adanerf_assemble_strips) on `tstream` right after it
    # has enqueued frame k+1.  No host sync anywhere in a step; flush() drains the last frame inside the timed region.
    tstreams = [torch.cuda.Stream(device=dev) for _ in rs]
    for q, ts in zip(rs, tstreams):
        q.set_stream(ts.cuda_stream)
    tstream = tstreams[0]
    n_buf = 2 if use_dist else 1
    M_pay = r.info.rays_local_max                 # payload rows of a (sub-)share: the same for every virtual rank
    outs = [torch.zeros((P, M_pay, 4), dtype=torch.uint8, device=dev) for _ in range(n_buf)]
    rgbs = [torch.zeros((max(q.info.rays_local, 1), 3), dtype=torch.float32, device=dev) for q in rs]
    rgb = rgbs[0]
    gathered = image = images = cstream = None
    ev_render = ev_gather = None
    # how the payloads travel (config.exchange.mode): gather = dist.gather on the data group; all_gather = all_gather_into_tensor on it (every
    # rank receives the 2.56 MB frame payload; the collective RCCL implements natively, gather being grouped send / recv); gloo_staged = device
    # -> host, gloo gather, host -> device (synchronous: the fallback of last resort, and `--exchange gloo`)
    xmode = {"rccl": "gather", "all_gather": "all_gather", "gloo": "gather"}[args.exchange]
    if use_dist and backend == "nccl" and data_pg is None:
        xmode = "gloo_staged"
    if use_dist:
        cstream = torch.cuda.Stream(device=dev)
        ev_render = [[torch.cuda.Event() for _ in range(P)] for _ in range(2)]
        ev_gather = [torch.cuda.Event() for _ in range(2)]
        gather_lists = [None, None]
        # (every rank holds the receive buffers: all_gather needs them everywhere, and a fallback may switch modes after they are made)
        gathered = [torch.zeros((world, P, M_pay, 4), dtype=torch.uint8, device=dev) for _ in range(2)]      # = [virtual rank][row][4]
        if rank == 0:
            gather_lists = [list(g.unbind(0)) for g in gathered]
            images = [torch.zeros((h * w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]      # one per buffer: two frames may be assembling
            image = images[0]
    state = {"k": 0, "pending": None, "gathers": 0, "last": 0, "exchange": True, "mode": xmode}

    # test hooks (tests/test_gpu_parity.py; never set by the driver): modes listed in ADANERF_BENCH_FAIL_MODES raise like a refusing RCCL would;
    # the rank named by ADANERF_BENCH_HANG_RANK stops for good in front of the timed phase, like a collective that never returns
    fail_modes = set(filter(None, os.environ.get("ADANERF_BENCH_FAIL_MODES", "").split(",")))
    hang_rank = int(os.environ.get("ADANERF_BENCH_HANG_RANK", "-1"))

    def exchange(b):
        """frame in buffer b: payloads of all ranks -> gathered[b] on rank 0 (enqueued on the current stream = cstream)"""
        mode = state["mode"]
        if mode in fail_modes:
            raise RuntimeError("exchange mode %s refused (ADANERF_BENCH_FAIL_MODES)" % mode)
        if mode == "gather":
            dist.gather(outs[b], gather_lists[b], dst=0, group=data_pg)
        elif mode == "all_gather":
            dist.all_gather_into_tensor(gathered[b].view(world * P * M_pay * 4), outs[b].view(P * M_pay * 4), group=data_pg)
        else:      # gloo_staged
            host = outs[b].cpu()                               # (synchronises cstream behind the renders it waited for)
            parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
            if rank == 0:
                gathered[b].copy_(torch.stack(parts), non_blocking=False)

    def finish(b):
        # frame in buffer b: its gather is complete -> de-interleave into the image on the stream of the context that rendered it
        ts, q = tstreams[b % fif], rs[b % fif]
        with torch.cuda.stream(ts):
            ts.wait_event(ev_gather[b])
            if rank == 0:
                q.assemble_strips(gathered[b].view(vworld, M_pay, 4), images[b])
        state["last"] = b

    poses = []
    if args.orbit > 0:
        size = np.array(scene["view_cell_size"], dtype=np.float32)
        for i in range(args.orbit):
            th = 2.0 * np.pi * i / args.orbit
            ppos = pose + 0.3 * 0.5 * size * np.array([np.cos(th), np.sin(th), 0.0], dtype=np.float32)
            prot = M.camera_rotation(100.0 + 360.0 * i / args.orbit, 0.0) if tag != "ndc_random_init" else rot
            poses.append((ppos.astype(np.float32), prot))

    def step():
        b = state["k"] & (n_buf - 1)
        # the contexts of this frame: all P sub-shares, or (two frames in flight) the one context buffer b belongs to
        lanes = list(range(len(rs))) if P > 1 else [b % fif]
        state["k"] += 1
        for k, i in enumerate(lanes):
            q, ts = rs[i], tstreams[i]
            if poses:
                q.set_camera(*poses[(state["k"] - 1) % len(poses)])
            with torch.cuda.stream(ts):
                if use_dist and state["k"] > 2:
                    ts.wait_event(ev_gather[b])             # frame k-2's payload has left this buffer
                q.render(outs[b][k], rgbs[i])
                if use_dist:
                    ev_render[b][k].record(ts)
        if use_dist and state["exchange"]:
            with torch.cuda.stream(cstream):
                for k in range(len(lanes)):
                    cstream.wait_event(ev_render[b][k])
                exchange(b)
                state["gathers"] += 1
                ev_gather[b].record(cstream)
            if state["pending"] is not None:
                finish(state["pending"])
            state["pending"] = b
        elif use_dist:                                           # render-only phase: nothing leaves the buffer
            with torch.cuda.stream(cstream):
                for k in range(len(lanes)):
                    cstream.wait_event(ev_render[b][k])
                ev_gather[b].record(cstream)

    def flush():
        if state["pending"] is not None:
            finish(state["pending"])
            state["pending"] = None

    def fence():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(n_steps):
        """n_steps steps between two fences; seconds, max over ranks (control plane)"""
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        flush()
        fence()
        dt_ = time.perf_counter() - t0
        if dist:
            t = torch.tensor([dt_], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_

    # Phase A (N > 1 only): the ranks' renders alone, no data-plane call at all -- what the GPUs do before any exchange is asked of them; the
    # value the line falls back to (flagged) if the exchange cannot run, and on a healthy run the cost of the exchange = phase B - phase A.
    render_only = None
    if use_dist:
        dog.phase = "phase A (render only)"
        state["exchange"] = False
        for _ in range(args.warmup):
            step()
        flush()
        fence()
        dt_a = timed(args.steps)
        render_only = {"value": args.steps / dt_a, "unit": "frames/s", "ms_per_step": dt_a / args.steps * 1e3,
                       "what": "every rank renders its share(s) of each frame, max over ranks; no exchange, no assembled image"}
        state["exchange"], state["k"] = True, 0
        if rank == 0:
            # what rank 0 prints if the exchange hangs: no N-GPU frames/s exists then (value null, exit status 3); phase A's rate under its own key
            dog.fallback = {"metric": "FPS at %dx%d" % (w, h), "value": None, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
                            "vs_baseline": None, "dtype": args.precision, "data": data, "render_only": render_only,
                            "config": {"workload": args.workload, "exchange": {"error": None, "errors": xerrors}}}
        # Pre-flight of the data plane: one exchange of the real payload in the chosen mode; a mode that raises on any rank is dropped for the
        # next one (gather -> all_gather -> gloo_staged), agreed over the control plane.  A mode that HANGS is the watchdog's business.
        while True:
            dog.phase = "exchange pre-flight (%s)" % state["mode"]
            err = None
            try:
                with torch.cuda.stream(cstream):
                    exchange(0)
                torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001
                err = "%s: %s: %s" % (state["mode"], type(e).__name__, str(e)[:300])
            if all_ok(err is None):
                break
            xerrors.append(err or "%s: failed on another rank" % state["mode"])
            nxt = {"gather": "all_gather", "all_gather": "gloo_staged"}.get(state["mode"])
            if nxt is None or (nxt == "all_gather" and backend != "nccl"):
                nxt = "gloo_staged" if state["mode"] != "gloo_staged" else None
            if nxt is None:
                raise SystemExit("bench.py: no exchange path works: %s" % "; ".join(xerrors))
            state["mode"] = nxt

    dog.phase = "warm-up"
    if use_dist and rank == hang_rank:
        time.sleep(1e6)
    for _ in range(args.warmup):
        step()
    flush()
    fence()
    for q in rs:
        q.set_profiling(True)
    dog.phase = "timed steps (exchange: %s)" % state["mode"]
    tele = GpuTelemetry(0 if os.environ.get("ADANERF_BENCH_ONE_DEVICE") == "1" else local_rank, period=0.005).start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    flush()
    fence()
    dt = time.perf_counter() - t0
    telemetry = tele.stop()
    st, frames = r.collect_stats()
    r.set_profiling(False)
    for q in rs[1:]:                              # the rank's other contexts (second frame in flight / sub-shares): same record, summed
        st2, f2 = q.collect_stats()
        q.set_profiling(False)
        if P == 1:
            frames += f2                          # (sub-shares render the same frames)
        for fld in ("total_samples", "batches", "ms_total", "ms_sample_mlp", "ms_compact", "ms_shade_mlp", "ms_composite", "shade_launches",
                    "sample_launches", "rays_refined"):
            setattr(st, fld, getattr(st, fld) + getattr(st2, fld))
        st.guard_max_seen = max(st.guard_max_seen, st2.guard_max_seen)
        st.guard_pair_seen = max(st.guard_pair_seen, st2.guard_pair_seen)
        for fld in ("guard_violations", "guard_audited", "guard_audit_mismatch", "guard_widened"):
            setattr(st, fld, getattr(st, fld) + getattr(st2, fld))
    r.lib.adanerf_get_info(r.handle, r.info)      # the guard band is calibrated at the first guarded frame
    exchange_rec = None
    if dist:
        on_rccl = backend == "nccl" and state["mode"] in ("gather", "all_gather")
        exchange_rec = {"mode": state["mode"], "backend": "nccl" if on_rccl else "gloo", "rccl_ranks": dist.get_world_size() if on_rccl else 0,
                        "control_plane": "gloo", "world_size": dist.get_world_size(), "gathers": state["gathers"],
                        "payload_bytes_per_rank": int(outs[0].numel()), "sub_shares_per_rank": P, "errors": xerrors, "error": None,
                        "render_only": render_only,
                        "exchange_cost_ms_per_frame": (dt / args.steps - render_only["ms_per_step"] * 1e-3) * 1e3 if render_only else None}
    if dist:
        dog.phase = "statistics (control plane)"
        dt_local = dt
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        if exchange_rec and render_only:
            exchange_rec["exchange_cost_ms_per_frame"] = (dt / args.steps) * 1e3 - render_only["ms_per_step"]
        # every rank's own numbers, so that ONE slow GPU of the node shows up as such: stage times, the clock and power its GPU ran the timed steps
        # at, and what bare MFMAs sustain on it (the in-process probe, ReLU-like operands)
        ptf, pmhz = (0.0, 0.0)
        if world > 1 and args.precision != "fp32" and not args.no_sustained_probe:
            dog.phase = "per-rank MFMA probe"
            ptf, pmhz = r.probe_mfma("relu", f16=(args.precision == "fp16"), target_ms=100.0)
            dog.phase = "statistics (control plane)"
        tot = torch.tensor([float(st.total_samples), float(st.ms_shade_mlp), float(st.ms_sample_mlp), float(telemetry.get("sclk_mhz_mean") or 0.0),
                            float(telemetry.get("power_w_mean") or 0.0), float(ptf), float(pmhz), float(dt_local)], dtype=torch.float64)
        tot_all = tot.clone()
        dist.all_reduce(tot_all, op=dist.ReduceOp.SUM)
        samples_all = float(tot_all[0].item())
        per_rank = [torch.zeros(8, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per_rank, tot)
        shard_samples = [float(x[0].item()) / max(frames, 1) for x in per_rank]
        shard_shade_ms = [float(x[1].item()) / max(frames, 1) for x in per_rank]
        shard_diag = {"sample_ms_per_frame": [float(x[2].item()) / max(frames, 1) for x in per_rank],
                      "sclk_mhz_mean": [float(x[3].item()) or None for x in per_rank], "power_w_mean": [float(x[4].item()) or None for x in per_rank],
                      "probe_relu_tflops": [float(x[5].item()) or None for x in per_rank], "probe_relu_clock_mhz": [float(x[6].item()) or None for x in per_rank],
                      "wall_ms_per_step": [float(x[7].item()) / args.steps * 1e3 for x in per_rank]}
    else:
        samples_all = float(st.total_samples)
        shard_samples = shard_shade_ms = shard_diag = None

    if rank == 0 and args.dump_image:
        np.save(args.dump_image, (images[state["last"]] if use_dist else outs[0][0][:h * w]).cpu().numpy().reshape(h, w, 4))

    ms_per_step = dt / args.steps * 1e3
    fps = args.steps / dt
    frames = max(frames, 1)
    samples_per_frame_local = st.total_samples / frames
    samples_per_frame = samples_all / frames
    mean_spp = samples_per_frame / (w * h)

    final_rec = None
    if rank == 0:
        # roofline of the dominant kernel (fused PE + shading MLP), this rank's launches
        launches = max(st.shade_launches, 1)
        shade_ms = st.ms_shade_mlp / launches
        flop_per_sample = shade_flop_per_sample(tag)
        flop_per_launch = flop_per_sample * (st.total_samples / launches)
        achieved = flop_per_launch / (shade_ms * 1e-3) / 1e12 if shade_ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.precision]
        kname = "shade_mlp%s_kernel" % ("32" if args.precision == "fp32" else ("16x2" if not generic_shape(tag) else "16_gen_staged"))
        if generic_shape(tag) and args.precision == "fp32":
            kname = "shade_mlp32_gen_kernel"
        # HBM bytes per launch of this kernel: PMC counters cannot be read from inside this process, so they come from the
        # committed rocprofv3 --pmc passes of this same command (tools/collect_profiles.sh -> profiles/).  A summary is used
        # only if it was taken with the very sources this library is built from (source_hash), else traffic is null.
        traffic, traffic_src = None, None
        import glob
        found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary_%s.json" % args.workload)))
        pmc_path = found[-1] if found else ""                 # the latest round's summary of this workload
        if found and world == 1 and r.info.batch_rays >= r.info.rays_local:
            try:
                pmc = json.load(open(pmc_path))
                meta = pmc.get("_meta", {})
                want = "--precision %s" % args.precision if args.precision != "bf16" else ""
                if meta.get("source_hash") == B.source_hash() and (want in meta.get("bench_args", "")) and \
                        ("--threshold" in meta.get("bench_args", "")) == (args.threshold is not None):
                    traffic = pmc.get(kname, {}).get("hbm_bytes_per_launch")
                    traffic_src = os.path.relpath(pmc_path, ROOT)
                else:
                    traffic_src = "stale: %s was collected with other sources / arguments" % os.path.relpath(pmc_path, ROOT)
            except Exception:
                traffic = None
        roofline = {"bound": "mfma", "kernel": kname,
                    "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                    "avg_launch_ms": shade_ms, "samples_per_launch": st.total_samples / launches,
                    "flop_per_sample": flop_per_sample}
        # What the silicon sustains on live operands, measured here and now on this box (adanerf_probe_mfma: register-only MFMA loops,
        # the shading kernel's wave shape): the part runs to a power budget, so the same instruction stream clocks ~2.4 GHz on zeros and
        # ~1.7 GHz on random operands (profiles/r04_mfma_peak_operands_clock.log).  `frac` stays against the nominal 2.5 PFLOP/s.
        if args.precision != "fp32" and not args.no_sustained_probe:
            sus = {}
            for mode, ms_t in (("zero", 60.0), ("relu", 200.0), ("random", 200.0)):
                tf, mhz = r.probe_mfma(mode, f16=(args.precision == "fp16"), target_ms=ms_t)
                sus[mode] = {"tflops": tf, "clock_mhz": mhz}
            roofline["sustained_peak"] = dict(sus, note="register-only v_mfma_f32_32x32x16 loops on every CU, same run, same box: operands all zero / "
                                              "random weights x post-ReLU-like activations (half zeros, rest positive) / random x random changing every MFMA")
            roofline["frac_of_sustained"] = achieved / sus["relu"]["tflops"] if sus["relu"]["tflops"] > 0 else None
            roofline["frac_of_sustained_random_operands"] = achieved / sus["random"]["tflops"] if sus["random"]["tflops"] > 0 else None
        roofline["box"] = telemetry      # shader clock and socket power of this rank's GPU over the timed steps (a slow box vs a slow kernel)
        stage_ms = {"sample_mlp": st.ms_sample_mlp / frames, "compact": st.ms_compact / frames,
                    "shade_mlp": st.ms_shade_mlp / frames, "composite": st.ms_composite / frames}
        smp_launch = max(st.sample_launches, 1)
        smp_tflops = SAMPLE_FLOP_PER_RAY * (sum(q.info.rays_local for q in (rs if P > 1 else rs[:1])) * frames / smp_launch) / (st.ms_sample_mlp / smp_launch * 1e-3) / 1e12 \
            if st.ms_sample_mlp > 0 else 0.0
        # Where the sampling stage sits against the MFMA roofline.  Algorithmic = 898 048 FLOP per ray; executed =
        # what the engine issues for it: the split engine three f16 MFMAs per term, the guarded mode one per term for every ray
        # plus three for each re-evaluated ray, the exact engine one on the fp32-MFMA pipe.
        R = sum(q.info.rays_local for q in (rs if P > 1 else rs[:1]))      # this rank's rays per frame
        refined = (st.rays_refined / frames) if args.sampling == "guarded" else 0.0
        exec_mult = {"split": 3.0, "fp16": 1.0, "fp32": 1.0, "guarded": 1.0 + 3.0 * refined / max(R, 1)}[args.sampling]
        smp_peak = PEAK_TFLOPS["fp32" if args.sampling == "fp32" else "fp16"]
        smp_flop = SAMPLE_FLOP_PER_RAY
        gen_fp32 = generic_wl and args.sampling == "fp32"
        if generic_wl:      # 6 x 128 oracle net: run-time-shaped kernels, split-precision pairs unless --sampling fp32 (no plain-fp16 pass)
            gl, gw, _ = generic_shape(tag)
            smp_flop = 2 * (90 * gw + (gl - 2) * gw * gw + gw * 128)
            smp_tflops *= smp_flop / SAMPLE_FLOP_PER_RAY
            exec_mult, smp_peak = (1.0, PEAK_TFLOPS["fp32"]) if gen_fp32 else (3.0, PEAK_TFLOPS["fp16"])
        smp_ms = st.ms_sample_mlp / frames
        sampling_roofline = {"bound": "mfma", "stage": "ray generation + encoding + sampling MLP (+ fused selection; guarded: + list + refinement pass)",
                             "engine": ("fp32 (run-time-shaped)" if gen_fp32 else "split-fp16 (run-time-shaped)") if generic_wl else args.sampling, "avg_ms_per_frame": smp_ms, "flop_per_ray": smp_flop,
                             "achieved_algorithmic": smp_tflops, "achieved_executed": smp_tflops * exec_mult, "peak": smp_peak, "unit": "TFLOP/s",
                             "frac_algorithmic": smp_tflops / smp_peak, "frac_executed": smp_tflops * exec_mult / smp_peak,
                             "rays_refined_per_frame": refined if args.sampling == "guarded" else None}
        if args.workload == "nerf_coarse_fine":
            # vanilla NeRF: the "sampling" stage is the COARSE network's pass -- uniform samples, the coarse 8 x 256 NeRF on the 16-bit shading engine,
            # classic compositing for the weights, the fine inverse-CDF sampler -- not a sampling MLP (round 5 mislabelled it: VERDICT weak 10)
            nc = int(r.info.num_samples_coarse)
            c_flop = nc * 1186816
            c_tflops = c_flop * R / (smp_ms * 1e-3) / 1e12 if smp_ms > 0 else 0.0
            sampling_roofline = {"bound": "mfma", "stage": "vanilla NeRF coarse pass: %d uniform samples per ray through the coarse 8x256 NeRF (%s MFMA, the shading engine) + "
                                                           "classic compositing + fine inverse-CDF sampler" % (nc, args.precision),
                                 "engine": "shade_mlp16x2_kernel (%s)" % args.precision, "avg_ms_per_frame": smp_ms, "flop_per_ray": c_flop, "flop_per_coarse_sample": 1186816,
                                 "achieved_algorithmic": c_tflops, "achieved_executed": c_tflops, "peak": PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
                                 "frac_algorithmic": c_tflops / PEAK_TFLOPS[args.precision], "frac_executed": c_tflops / PEAK_TFLOPS[args.precision],
                                 "rays_refined_per_frame": None}
            smp_tflops = c_tflops
        # HBM-side view of the two bandwidth-bound stages: bytes the stage's kernels move by construction
        S_loc = samples_per_frame_local
        fused = args.sampling in ("split", "fp16", "guarded") and 0.0 < thr and n_max <= 16 and args.workload != "nerf_coarse_fine" and \
            not (generic_wl and args.sampling == "fp32")      # (the run-time-shaped split-precision sampling kernel fuses it too)
        if args.workload == "nerf_coarse_fine":
            comp_bytes, comp_what = None, "fine sampler (not an HBM-bound stage)"
        elif thr == 0.0:      # dense: keys are implicit and the oracle buffer is the weight array; only offsets + counts are written
            comp_bytes, comp_what = R * 8, "dense_offsets_kernel: offsets + counts out (a latency-sized launch, not a bandwidth measurement)"
        elif fused:      # selection ran inside the sampling kernel: the stage is expand_kernel alone
            comp_bytes = R * 4 + S_loc * 5 + R * 4 + S_loc * 8
            comp_what = "expand_kernel: counts + kept (bin, value) rows in, offsets + keys + weights out"
        else:            # selection kernel over the [R,128] oracle buffer, then expand_kernel
            comp_bytes = R * 512 + R * 4 + S_loc * 5 + R * 4 + S_loc * 5 + R * 4 + S_loc * 8
            comp_what = "selection kernel over the oracle buffer + expand_kernel"
        cmp_bytes = S_loc * 20 + R * 8 + R * 16
        comp_gbps = comp_bytes / (stage_ms["compact"] * 1e-3) / 1e9 if comp_bytes and stage_ms["compact"] > 0 else None
        cmp_gbps = cmp_bytes / (stage_ms["composite"] * 1e-3) / 1e9 if stage_ms["composite"] > 0 else None
        hbm = {"compact_GBps": comp_gbps, "compact_frac": comp_gbps / 8000.0 if comp_gbps else None, "compact_bytes_per_frame": comp_bytes,
               "compact_stage": comp_what, "composite_GBps": cmp_gbps, "composite_frac": cmp_gbps / 8000.0 if cmp_gbps else None,
               "composite_bytes_per_frame": cmp_bytes, "peak_GBps": 8000.0}
        assert not comp_gbps or comp_gbps < 8000.0, "compact stage above the HBM peak: the byte model is wrong"

        cpu = None
        quality = {}
        if not args.no_cpu_baseline and world == 1:
            dog.phase = "extra: CPU baseline"
            cpu, ref, (row0, rows), psnr = cpu_baseline(td, w, h, pose, rot, args.cpu_budget)
            if poses:                                   # the quality check refers to the fixed pose
                r.set_camera(pose, rot)
                with torch.cuda.stream(tstream):
                    r.render(outs[0][0], rgb)
                torch.cuda.synchronize()
            mine = rgb.cpu().numpy()[ref["sel"]]
            cnt = r.buffer(3, np.int32, (r.info.rays_local,))[ref["sel"]] if r.info.batch_rays >= r.info.rays_local else None
            if cnt is not None:
                same = cnt == ref["count"]
                quality = {"psnr_vs_oracle_db": psnr(mine[same], ref["rgb"][same]),
                           "max_abs_err_vs_oracle": float(np.abs(mine[same] - ref["rgb"][same]).max()),
                           "rays_with_identical_sample_count": float(same.mean()), "rays_checked": int(same.size)}
        # opt-in speed mode, reported beside the headline (never as it): the same frame with the sampling MLP in plain
        # fp16 (the viewer's TensorRT arithmetic); selection then deviates from the fp32 path on ~1 % of rays
        speed = None
        dog.phase = "extra: speed / exact / guarded / split-frame modes"
        if world == 1 and args.sampling in ("split", "guarded") and not args.no_speed_mode and not generic_wl:
            with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h, batch_size=args.batch_rays), precision=args.precision,
                                           sampling="fp16", device_id=local_rank) as r2:
                r2.set_camera(pose, rot)
                rgb2 = r2.empty((w * h, 3), np.float32)
                for _ in range(args.warmup):
                    r2.render(None, rgb2)
                r2.sync()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    r2.render(None, rgb2)
                r2.sync()
                dt2 = time.perf_counter() - t1
                st2 = r2.render(None, rgb2, stats=True)
                speed = {"sampling": "plain fp16 (ADANERF_SAMPLING_FP16)", "value": args.steps / dt2, "unit": "frames/s",
                         "sample_mlp_ms": st2.ms_sample_mlp, "mean_samples_per_ray": st2.total_samples / float(w * h)}
                if cpu is not None:
                    mine2 = rgb2.numpy()[ref["sel"]]
                    cnt2 = r2.buffer(3, np.int32, (w * h,))[ref["sel"]] if r2.info.batch_rays >= w * h else None
                    speed["psnr_vs_oracle_db_all_rays"] = psnr(mine2, ref["rgb"])
                    if cnt2 is not None:
                        speed["rays_with_identical_sample_count"] = float((cnt2 == ref["count"]).mean())
        # the same frame with the split-precision engine on EVERY ray: exact by construction, no band, no audit -- printed next to
        # the headline so that the cost of that guarantee is on the line
        exact = None
        if world == 1 and args.sampling == "guarded" and not args.no_exact_mode and not generic_wl:
            with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h, batch_size=args.batch_rays), precision=args.precision,
                                           sampling="split", device_id=local_rank) as r3:
                r3.set_camera(pose, rot)
                out3 = r3.empty((w * h, 4), np.uint8)
                for _ in range(args.warmup):
                    r3.render(out3, None)
                r3.sync()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    r3.render(out3, None)
                r3.sync()
                dt3 = time.perf_counter() - t1
                st3 = r3.render(out3, None, stats=True)
                cnt3 = r3.buffer(3, np.int32, (w * h,)) if r3.info.batch_rays >= w * h else None
                cnt1 = r.buffer(3, np.int32, (w * h,)) if (cnt3 is not None and not poses) else None
                exact = {"sampling": "split-fp16 on every ray (ADANERF_SAMPLING_SPLIT_FP16)", "value": args.steps / dt3, "unit": "frames/s",
                         "sample_mlp_ms": st3.ms_sample_mlp, "samples_per_frame": int(st3.total_samples),
                         "rays_with_the_headline_modes_sample_count": float((cnt1 == cnt3).mean()) if cnt1 is not None else None}
        # the opt-in guarded two-precision selection on the same frame, same box, same run: what the default rule compares.  Its band,
        # monitor and audit counters ride along; the selection must be the headline's on every ray.
        guarded = None
        if world == 1 and args.sampling == "split" and args.precision != "fp32" and not args.no_guarded_mode and not generic_wl and 0.0 < thr and n_max <= 16 \
                and args.workload != "nerf_coarse_fine":
            with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h, batch_size=args.batch_rays), precision=args.precision, sampling="guarded",
                                           guard_eps=args.guard_eps, guard_audit_period=args.guard_audit_period, device_id=local_rank) as r4:
                r4.set_camera(pose, rot)
                out4 = r4.empty((w * h, 4), np.uint8)
                for _ in range(args.warmup):
                    r4.render(out4, None)
                r4.sync()
                dt4g = None
                for _ in range(3):      # three short passes, the fastest counts: a side measurement of 30 frames is at the mercy of one hiccup
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        r4.render(out4, None)
                    r4.sync()
                    dt4g = min(dt4g or 1e30, time.perf_counter() - t1)
                st4 = r4.render(out4, None, stats=True)
                r4.lib.adanerf_get_info(r4.handle, r4.info)
                cnt4 = r4.buffer(3, np.int32, (w * h,)) if r4.info.batch_rays >= w * h else None
                cnt1 = r.buffer(3, np.int32, (w * h,)) if (cnt4 is not None and not poses) else None
                g_fps = args.steps / dt4g
                guarded = {"sampling": "guarded two-precision (ADANERF_SAMPLING_GUARDED): plain fp16 on every ray, split-fp16 on the rays inside the audited band",
                           "value": g_fps, "unit": "frames/s", "passes": "fastest of 3 x %d frames" % args.steps, "ahead_of_the_headline": g_fps / fps - 1.0,
                           "default_rule": "per workload: the library default (and this line's value) is the mode that is exact by construction; a host started with "
                                           "--sampling auto (adanerf_amd.choose_sampling) measures both on its own workload and takes this mode when it is >= 8 %% "
                                           "ahead -- on this workload it %s" % ("is: auto picks guarded" if g_fps / fps - 1.0 >= 0.08 else "is not: auto stays with split"),
                           "auto_choice": "guarded" if g_fps / fps - 1.0 >= 0.08 else "split",
                           "sample_mlp_ms": st4.ms_sample_mlp, "rays_refined": int(st4.rays_refined), "samples_per_frame": int(st4.total_samples),
                           "rays_with_the_headline_modes_sample_count": float((cnt1 == cnt4).mean()) if cnt1 is not None else None,
                           "guard": {"eps": float(r4.info.guard_eps), "eps_pair": float(r4.info.guard_eps_pair),
                                     "band_source": R_GUARD_FROM.get(int(r4.info.guard_calib_source)), "calibration_poses": int(r4.info.guard_calib_poses),
                                     "monitor_max_seen": float(st4.guard_max_seen), "monitor_pair_seen": float(st4.guard_pair_seen),
                                     "monitor_violations": int(st4.guard_violations), "band_widened": int(st4.guard_widened),
                                     "audit_period": int(r4.info.guard_audit_period), "rays_audited": int(st4.guard_audited),
                                     "audit_mismatches": int(st4.guard_audit_mismatch),
                                     "note": "monitor / audit counters are cumulative since the context was created (warm-up included)"}}
        # the same frame as two concurrent sub-shares (virtual ranks 0 and 1 of a world of 2) on two contexts / streams of this one GPU,
        # assembled by adanerf_assemble_strips: what `--gpus N` does per rank (P above), measured at N = 1.  Reported beside the
        # headline: the headline renders one context at a time so that its per-stage times are those of the kernels alone.
        split = None
        if world == 1 and not use_dist and not args.no_split_mode and args.batch_rays <= 0:
            sr2 = sharding.balanced_strip_rows(h, 2)
            qs = [adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, w, h), precision=args.precision, sampling=args.sampling, guard_eps=args.guard_eps,
                                             guard_audit_period=args.guard_audit_period, device_id=local_rank,
                                             shard_rank=k, shard_world=2, strip_rows=sr2) for k in range(2)]
            for q in qs:
                q.init()
                q.set_camera(pose, rot)
            ss = [torch.cuda.Stream(device=dev) for _ in qs]
            for q, s_ in zip(qs, ss):
                q.set_stream(s_.cuda_stream)
            pay = torch.zeros((2, qs[0].info.rays_local_max, 4), dtype=torch.uint8, device=dev)
            img = torch.zeros((h * w, 4), dtype=torch.uint8, device=dev)
            ev = torch.cuda.Event()

            def split_step():
                for k in (1, 0):
                    with torch.cuda.stream(ss[k]):
                        qs[k].render(pay[k], None)
                        if k == 1:
                            ev.record(ss[1])
                with torch.cuda.stream(ss[0]):
                    ss[0].wait_event(ev)
                    qs[0].assemble_strips(pay, img)
                    ss[1].wait_stream(ss[0])          # the next frame's sub-share 1 must not overwrite a payload being assembled

            for _ in range(args.warmup):
                split_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                split_step()
            torch.cuda.synchronize()
            dt4 = time.perf_counter() - t1
            lat = []
            for _ in range(10):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                split_step()
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t1)
            same_img = None
            if not poses:
                with torch.cuda.stream(tstream):
                    r.render(outs[0][0], rgb)
                torch.cuda.synchronize()
                same_img = bool(torch.equal(img, outs[0][0][:h * w]))
            split = {"what": "the frame as 2 concurrent sub-shares on 2 contexts / streams + adanerf_assemble_strips (as every rank of --gpus N does)",
                     "value": args.steps / dt4, "unit": "frames/s", "frame_latency_ms_median": float(np.median(lat)) * 1e3,
                     "image_identical_to_the_headline_frame": same_img}
            for q in qs:
                q.close()
        rec = {"metric": "FPS at %dx%d" % (w, h), "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": args.precision, "data": data,
               "config": {"workload": ("nerf_coarse_fine: %dx%d, vanilla NeRF, %d uniform coarse samples + %d fine samples per ray (%d through the fine network), coarse and fine "
                                       "8x256 NeRF %s, classic compositing; no sampling MLP" % (w, h, int(r.info.num_samples_coarse), n_max, int(r.info.num_samples_coarse) + n_max, args.precision))
                                      if args.workload == "nerf_coarse_fine" else "%s: %dx%d, N=%d, threshold %.2f, %s shading MLP %s, sampling MLP %s" %
                                      (args.workload, w, h, n_max, thr, args.workload[8:] if generic_wl else "8x256", args.precision,
                                       ("%s on the run-time-shaped %s kernel" % (args.workload[8:], "exact-fp32" if args.sampling == "fp32" else "split-fp16")) if generic_wl else
                                       {"split": "split-fp16 (3 MFMAs per term)", "fp32": "fp32 MFMA", "fp16": "plain fp16 (opt-in speed mode)",
                                        "guarded": "guarded two-precision (plain fp16, split-fp16 on the rays inside the band)"}[args.sampling]),
                          "parallelism": ("image-strip shard x%d (%d-row strips, round-robin over %d virtual ranks, %d concurrent sub-share(s) per GPU) + %s gather "
                                          "overlapped with the next frame" % (world, strip_rows, vworld, P, {"gather": "RCCL" if backend == "nccl" else backend, "all_gather": "RCCL all_gather",
                                                                                                  "gloo_staged": "host-staged gloo"}[state["mode"]])) if use_dist else "single GPU",
                          "exchange": exchange_rec,
                          "frames_in_flight": fif, "sub_shares_per_gpu": P,
                          "rays_refined_per_frame": (st.rays_refined / frames) if args.sampling == "guarded" else None,
                          "guard": ({"eps": float(r.info.guard_eps), "eps_pair": float(r.info.guard_eps_pair),
                                     "band_source": R_GUARD_FROM.get(int(r.info.guard_calib_source)), "calibration_poses": int(r.info.guard_calib_poses),
                                     "monitor_max_seen": float(st.guard_max_seen), "monitor_pair_seen": float(st.guard_pair_seen),
                                     "monitor_violations": int(st.guard_violations), "band_widened": int(st.guard_widened),
                                     "audit_period": int(r.info.guard_audit_period),
                                     "audit_fill": "the audit fills the refinement pass's last round (at least a quarter of the 1 / period quota per frame)",
                                     "rays_audited": int(st.guard_audited),
                                     "audit_mismatches": int(st.guard_audit_mismatch),
                                     "note": "monitor / audit counters are cumulative since the context was created (warm-up included)"}
                                    if args.sampling == "guarded" else None),
                          "batch_rays": r.info.batch_rays, "mean_samples_per_ray": mean_spp, "samples_per_frame": samples_per_frame,
                          "camera": ("%d-pose orbit inside the view cell" % args.orbit) if poses else "fixed: view-cell centre, yaw 100 deg"},
               "roofline": roofline, "cpu_baseline": cpu, "stage_ms_per_frame": stage_ms,
               "sampling_mlp_algorithmic_tflops": smp_tflops, "sampling_roofline": sampling_roofline, "hbm_stages": hbm, "quality": quality, "guarded_mode": guarded, "exact_mode": exact, "speed_mode": speed,
               "split_frame_mode": split}
        if P > 1 or fif > 1:
            rec["stage_ms_note"] = ("rank 0's contexts run concurrently on their own streams (%s): the per-stage HIP-event times are summed over them "
                                    "and overlap in time, so they add up to more than ms_per_step" % ("%d sub-shares of the frame" % P if P > 1 else "two frames in flight"))
        if shard_samples:
            mean_s = sum(shard_samples) / len(shard_samples)
            rec["shards"] = dict({"samples_per_frame": shard_samples, "shade_ms_per_frame": shard_shade_ms,
                                  "sample_imbalance_max_over_mean": max(shard_samples) / mean_s if mean_s > 0 else None}, **(shard_diag or {}))
        final_rec = dog.record = rec                    # from here on a hang costs the extras below, not the line
    # N > 1: short measurements of the other exchange paths, after the line's own numbers are final.  (1) the other RCCL collective, in this
    # process group; (2) --exchange peer in a child process of rank 0 -- one process driving all N GPUs with hipMemcpyPeerAsync -- while the
    # other ranks wait at a control-plane barrier with their GPUs idle.  Each is optional: whatever fails is recorded as text.
    if dist and world > 1 and not args.no_alternatives:
        alts = {}
        k_alt = max(2, min(args.steps, 10))
        main_mode = state["mode"]
        for mode in (("all_gather",) if main_mode == "gather" else ("gather",) if main_mode == "all_gather" else ()):
            if backend != "nccl" and mode == "all_gather" and os.environ.get("ADANERF_BENCH_ALT_ALL_GATHER") != "1":
                continue                                  # (gloo has no all_gather_into_tensor for device tensors worth timing)
            dog.phase = "alternative exchange (%s)" % mode
            state["mode"], state["k"], err = mode, 0, None
            try:
                with torch.cuda.stream(cstream):
                    exchange(0)
                torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, str(e)[:300])
            if not all_ok(err is None):
                alts[mode] = {"error": err or "failed on another rank"}
                continue
            for _ in range(2):
                step()
            flush()
            fence()
            dt_alt = timed(k_alt)
            alts[mode] = {"value": k_alt / dt_alt, "unit": "frames/s", "ms_per_step": dt_alt / k_alt * 1e3, "steps": k_alt}
        state["mode"] = main_mode
        dog.phase = "alternative exchange (peer, child process of rank 0)"
        dist.barrier()                                    # every rank's GPU is idle from here
        if rank == 0:
            import subprocess
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                    "ROLE_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--exchange", "peer", "--steps", str(max(2, min(args.steps, 20))),
                   "--warmup", "3", "--workload", args.workload, "--precision", args.precision, "--sampling", args.sampling,
                   "--sub-shares", str(P), "--batch-rays", str(args.batch_rays)] + (["--threshold", str(args.threshold)] if args.threshold is not None else [])
            try:
                out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
                if out.returncode == 0 and line:
                    pr = json.loads(line[-1])
                    alts["peer"] = {"value": pr["value"], "unit": "frames/s", "ms_per_step": pr["ms_per_step"], "steps": pr["steps"],
                                    "what": pr["config"]["parallelism"]}
                else:
                    alts["peer"] = {"error": "rc %d: %s" % (out.returncode, (out.stderr or out.stdout)[-400:])}
            except Exception as e:      # noqa: BLE001
                alts["peer"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        dist.barrier()
        if rank == 0:
            final_rec["config"]["exchange"]["alternatives"] = alts
    if rank == 0:
        dog.cancel()
        print(json.dumps(final_rec))
        sys.stdout.flush()
    dog.cancel()
    for q in rs:
        q.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
