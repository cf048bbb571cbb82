"""Which samples of a shade_mlp launch disagree with the oracle (debugging aid for kernel variants): prints the wrong
sample ranges as (tile of 256, wave of 32) pairs.  ADANERF_LIB selects the library."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import adanerf_oracle as O
from conftest import case_weights, load_case
from test_gpu_parity import golden_ray_records, golden_samples
import adanerf_amd

z, meta, sc = load_case("classroom_n8_thr02")
wts = case_weights(meta)
td = tempfile.mkdtemp()
O.write_model_dir(td, sc, wts)
count, off, key, sw, sray, sbin = golden_samples(z, sc)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for S in (6000, key.shape[0]):
    zt = O.bin_t(sbin[:S].astype(np.int64))
    feat = O.shading_inputs(z["p"], z["nds"], sray[:S], O.to_world_depth(zt, sc), sc, meta["w"], meta["h"])
    ref = O.shading_mlp(feat, wts.net1, 63)
    for prec in ("fp16", "bf16"):
        with adanerf_amd.NeuralRenderer(adanerf_amd.Settings(td, meta["w"], meta["h"]), precision=prec) as r:
            r.set_camera(z["pose"], z["rot"])
            d_rays = r.to_device(golden_ray_records(z, meta, sc))
            d_key = r.to_device(key[:S])
            d_tot = r.to_device(np.array([S], dtype=np.int32))
            raw = r.empty((S, 4), np.float32)
            for it in range(reps):
                r.shade_mlp(d_rays, d_key, d_tot, S, raw)
                out = raw.numpy()
                bad = np.abs(out - ref).max(axis=1) > (0.06 + 0.02 * np.abs(ref).max(axis=1)) * (4 if prec == "bf16" else 1)
                idx = np.nonzero(bad)[0]
                waves = sorted(set((int(i) // 256, (int(i) % 256) // 32) for i in idx))
                print("S=%d %s run %d: %d wrong samples; (tile, wave): %s" % (S, prec, it, idx.size, waves[:40]))
                if it == 0 and S == 6000:
                    err = np.abs(out - ref)
                    w47 = ((np.arange(S) % 256) // 32) >= 4
                    print("   rms err waves 0-3: %.5f  waves 4-7: %.5f; per channel (4-7): %s" %
                          (np.sqrt((err[~w47] ** 2).mean()), np.sqrt((err[w47] ** 2).mean()), np.sqrt((err[w47] ** 2).mean(0)).round(5)))
                    print("   lanes (sample %% 32) of wrong samples:", np.bincount(idx % 32, minlength=32).tolist())
                    print("   first wrong samples:", [(int(i), out[i].round(3).tolist(), ref[i].round(3).tolist()) for i in idx[:4]])
