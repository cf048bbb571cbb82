"""The committed fixtures are what oracle/gen_golden.py produces from the reference TODAY: where /root/reference exists
(the build container; never the GPU box) two cases are regenerated into a scratch directory and compared with the
committed files array by array, bit for bit.  Guards against fixtures drifting behind their generator."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT

REGEN = ["classroom_n8_thr02", "ndc_synthetic_n8", "ray_table_ref"]      # ray_table_ref: the reference's own pixel-ray table (round 5)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference is only present in the build container")
def test_committed_fixtures_equal_a_fresh_generator_run(tmp_path):
    out = str(tmp_path)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), "--only", ",".join(REGEN), "--out", out],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    for name in REGEN:
        new, old = np.load(os.path.join(out, name + ".npz")), np.load(os.path.join(GOLD, name + ".npz"))
        assert sorted(new.files) == sorted(old.files), (name, sorted(set(new.files) ^ set(old.files)))
        for k in new.files:
            if k == "meta" and name != "ray_table_ref":
                assert json.loads(bytes(new[k]).decode()) == json.loads(bytes(old[k]).decode()), name
            else:
                assert new[k].dtype == old[k].dtype and new[k].shape == old[k].shape, (name, k)
                assert np.array_equal(new[k], old[k], equal_nan=new[k].dtype.kind == "f"), (name, k)
