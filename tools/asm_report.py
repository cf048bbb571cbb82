#!/usr/bin/env python3
"""Compiles the device code to gfx950 assembly and reports, for kernels matching a regex, the instruction
histogram and where scratch (spill) traffic sits relative to the MFMA stream."""
import bisect, collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/adanerf_all.s"
sys.path.insert(0, root)
from adanerf_amd import build as B
parts = []
for src in ("adanerf_hip.hip", "launch_f32.hip"):          # same per-translation-unit flags as the shipped build
    tmp = "/tmp/adanerf_%s.s" % src
    subprocess.run(["hipcc"] + B.HIPCC_FLAGS + B.TU_FLAGS.get(src, []) + ["-S", "--cuda-device-only", os.path.join(root, "adanerf_amd/csrc", src), "-o", tmp],
                   check=True, stderr=subprocess.DEVNULL)
    parts.append(open(tmp).read())
open(out, "w").write("\n".join(parts))
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 18
for f in re.split(r"\n\s*\.globl\s+", open(out).read()):
    name = f.split("\n", 1)[0].strip()
    if not pat.search(name) or "v_mfma" not in f and "ds_" not in f:
        continue
    lines = f.split("\n")
    c = collections.Counter(m.group(1) for l in lines for m in [re.match(r"\s+([a-z_0-9]+)", l)] if m)
    print("==", name, len(lines), "lines")
    print("  " + ", ".join("%s:%d" % kv for kv in c.most_common(top)))
    mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
    sc = [i for i, l in enumerate(lines) if "scratch_" in l]
    if mf:
        h = collections.Counter(bisect.bisect(mf, i) // 128 for i in sc)
        print("  mfma:%d scratch:%d per-128-mfma-group:%s" % (len(mf), len(sc), sorted(h.items())))
        vm0 = [i for i, l in enumerate(lines) if "s_waitcnt" in l and "vmcnt(0)" in l and mf[0] < i < mf[-1]]
        print("  vmcnt(0) waits inside the mfma range:", len(vm0))
