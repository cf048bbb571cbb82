#!/usr/bin/env python3
"""One worker of bench.py's cpu_baseline leg: renders a band of image rows with the numpy/torch-GEMM oracle
(adanerf_oracle.render_frame, the restatement of the reference's PyTorch CPU path) on `threads` host threads.
TEST / BASELINE INFRASTRUCTURE ONLY -- never part of the product path.

  cpu_worker.py <model_dir> <w> <h> <row0> <rows> <threads> <budget_s> <sync_dir> <worker_id> <pose/rot .npy>

Protocol: import + warm-up, touch <sync_dir>/ready.<id>, wait for <sync_dir>/go (so that all workers compute at the same
time and share the machine the way one parallel render would), then render its rows in slices until the rows are done or
the time budget is used up; writes <sync_dir>/out.<id>.npz with rows_done, seconds and the band's rgb / sample counts."""
import os
import sys
import time

model_dir, w, h, row0, rows, threads, budget, sync_dir, wid, cam = sys.argv[1:11]
w, h, row0, rows, threads, budget = int(w), int(h), int(row0), int(rows), int(threads), float(budget)
for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ[k] = str(threads)

import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import adanerf_oracle as O  # noqa: E402

backend = "numpy"
try:
    import torch
    torch.set_num_threads(threads)
    O.set_matmul_backend("torch")
    backend = "torch CPU GEMM"
except Exception:
    O.set_matmul_backend("numpy")

sc = O.load_scene(model_dir)
wts = O.load_weights(model_dir)
c = np.load(cam)
pose, rot = c[:3].astype(np.float32), c[3:].reshape(3, 3).astype(np.float32)
CH = 16384
O.render_frame(sc, wts, w, h, pose, rot, chunk=CH, rows=(row0, row0 + 1))      # warm-up (thread pools, page cache)
open(os.path.join(sync_dir, "ready.%s" % wid), "w").close()
go = os.path.join(sync_dir, "go")
while not os.path.exists(go):
    time.sleep(0.005)
t0 = time.time()
step = max(1, CH // w)
done = 0
rgb, cnt = [], []
while done < rows:
    n = min(step, rows - done)
    r = O.render_frame(sc, wts, w, h, pose, rot, chunk=CH, rows=(row0 + done, row0 + done + n))
    rgb.append(r["rgb"])
    cnt.append(r["count"])
    done += n
    if time.time() - t0 > budget:
        break
dt = time.time() - t0
np.savez(os.path.join(sync_dir, "out.%s.npz" % wid), rows_done=done, seconds=dt, row0=row0, rgb=np.concatenate(rgb),
         count=np.concatenate(cnt), backend=np.array(backend))
