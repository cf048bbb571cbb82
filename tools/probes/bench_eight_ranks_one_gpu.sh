#!/bin/bash
# bench.py launched exactly as the driver launches it for N = 8, with all eight ranks on this box's one GPU and gloo for the exchange
# (RCCL refuses two ranks on one device): checks the one JSON line and that the assembled frame equals the single-rank frame.
cd "$(dirname "$0")/../.."
O=gpurun_out
python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-speed-mode --dump-image /tmp/one.npy > $O/r03_bench_1rank.json 2>/dev/null
ADANERF_BENCH_DIST_BACKEND=gloo ADANERF_BENCH_ONE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
  --master-port 29761 bench.py --gpus 8 --steps 6 --warmup 2 --no-cpu-baseline --dump-image /tmp/eight.npy > $O/r03_bench_8ranks_one_gpu.json 2> $O/r03_bench_8ranks_one_gpu.err
python - <<'PY'
import json, numpy as np
a, b = np.load("/tmp/one.npy"), np.load("/tmp/eight.npy")
r = json.loads([l for l in open("gpurun_out/r03_bench_8ranks_one_gpu.json") if l.startswith("{")][-1])
print("frames equal:", bool(np.array_equal(a, b)), "| n_gpus", r["n_gpus"], "value", round(r["value"], 1), "frames_in_flight", r["config"]["frames_in_flight"],
      "exchange", r["config"]["exchange"], "| shards", {k: (v if not isinstance(v, list) else [round(x, 3) for x in v]) for k, v in r.get("shards", {}).items()})
PY
tail -3 $O/r03_bench_8ranks_one_gpu.err
