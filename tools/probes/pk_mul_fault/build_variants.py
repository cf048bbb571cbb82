#!/usr/bin/env python3
"""Builds the out-of-tree libraries of the round-3 reproducibility fault (README.md) into tools/ablate_libs/pkf_<name>.so:
a copy of adanerf_amd/csrc with composite_wave_kernel replaced by its round-3 shuffle form (composite_wave_r03.inc), compiled with the
shipped per-translation-unit flags.  The shipped sources and their hash are not touched.

    python tools/probes/pk_mul_fault/build_variants.py            # v0, v24, v26, v0_noslp, v26_noslp
    bash tools/probes/pk_mul_fault/run.sh                         # on the GPU box
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
from adanerf_amd import build as B   # noqa: E402

VARIANTS = {"v0": ["-DPKF_VARIANT=0"], "v24": ["-DPKF_VARIANT=24"], "v26": ["-DPKF_VARIANT=26"],
            "v0_noslp": ["-DPKF_VARIANT=0", "-fno-slp-vectorize"], "v26_noslp": ["-DPKF_VARIANT=26", "-fno-slp-vectorize"]}


def patched_csrc():
    tmp = tempfile.mkdtemp(prefix="pkf_csrc_")
    dst = os.path.join(tmp, "adanerf_amd", "csrc")
    shutil.copytree(B.CSRC, dst)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    shutil.copy(os.path.join(HERE, "composite_wave_r03.inc"), dst)
    p = os.path.join(dst, "k_composite.hip.hpp")
    s = open(p).read()
    a = s.index("__global__ __launch_bounds__(256) void composite_wave_kernel(")
    b = s.index("// multi-GPU: gathered [world][rays_local_max] uchar4")
    open(p, "w").write(s[:a] + '#include "composite_wave_r03.inc"\n\n' + s[b:])
    # the library's own rule "no lane exchange through the LDS crossbar" is a CPU test on the shipped sources, not a build check
    return dst


def main():
    names = sys.argv[1:] or list(VARIANTS)
    csrc = patched_csrc()
    B.CSRC = csrc
    outdir = os.path.join(ROOT, "tools", "ablate_libs")
    os.makedirs(outdir, exist_ok=True)
    for n in names:
        out = os.path.join(outdir, "pkf_%s.so" % n)
        B.build_library(force=True, out=out, extra_flags=VARIANTS[n])
        # the instruction the fault correlates with, in this build's composite_wave_kernel
        asm = subprocess.run([B._hipcc()] + B.HIPCC_FLAGS + B.TU_FLAGS["adanerf_hip.hip"] + VARIANTS[n] +
                             ["-S", "--cuda-device-only", os.path.join(csrc, "adanerf_hip.hip"), "-o", "-"], capture_output=True, text=True, cwd=csrc).stdout
        m = re.search(r"\n_ZN7adanerf21composite_wave_kernel.*?s_endpgm", asm, re.S)
        body = m.group(0) if m else ""
        open(os.path.join(outdir, "pkf_%s.composite_wave_kernel.s" % n), "w").write(body)
        print("%-10s %s  v_pk_mul_f32 %d  v_mul_f32 %d  ds_bpermute_b32 %d" % (n, out, body.count("v_pk_mul_f32"), len(re.findall(r"v_mul_f32", body)), body.count("ds_bpermute_b32")))


if __name__ == "__main__":
    main()
