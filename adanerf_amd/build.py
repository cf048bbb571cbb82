"""In-tree build of the HIP shared library and the host CLI (hipcc, gfx950 only)."""
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIBDIR = os.path.join(HERE, "lib")
BINDIR = os.path.join(HERE, "bin")
LIB_SOURCES = ["adanerf_hip.hip", "launch_f32.hip", "format.cpp", "pack.cpp"]
_INCLUDE = re.compile(r'^[ \t]*#[ \t]*include[ \t]+"([^"]+)"', re.M)


def include_closure(sources=None):
    """Every file the library's translation units reach through `#include "..."` (paths relative to csrc/, sources included),
    sorted.  This IS the dependency list: a header that is included but not listed here cannot exist, so `_stale()` rebuilds on
    any edit and `source_hash()` covers every kernel a profile may have been taken with (round 3 missed k_generic16.hip.hpp)."""
    seen, todo = set(), list(sources or LIB_SOURCES)
    while todo:
        rel = os.path.normpath(todo.pop())
        if rel in seen:
            continue
        path = os.path.join(CSRC, rel)
        if not os.path.exists(path):
            if os.path.basename(rel).startswith("x_"):      # experiment-only headers live in tools/experiments/ (reached through -I in
                continue                                    # -DADN_EXPERIMENT builds only: tools/ablate.sh); never part of the shipped library
            raise RuntimeError("%s: included by the library sources but missing" % path)
        seen.add(rel)
        for inc in _INCLUDE.findall(open(path, encoding="utf-8", errors="replace").read()):
            todo.append(os.path.join(os.path.dirname(rel), inc))
    return sorted(seen)


LIB_DEPS = include_closure()
ARCH = "gfx950"
# -ffp-contract=off: the fused and the debug kernels must generate bit-identical rays.
# -fno-slp-vectorize (round 5): no packed-fp32 VALU (v_pk_mul / add / fma_f32) anywhere in the device code.  The SLP vectoriser emitted ~4 300 of
# them; the only reproducibility fault this library has had (round 3, tools/probes/pk_mul_fault/) needs SLP-packed fp32 code around a packed
# product and disappears with this flag, and the flag costs nothing measurable (profiles/r05_variants_slp_splitpack.log).  tests/test_host_cpu.py
# checks the assembly.
HIPCC_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"]
# Per translation unit: the main one keeps MFMA accumulators in architectural VGPRs, so the VALU epilogues of the 16-bit
# kernels read them without v_accvgpr_read copies (split-precision sampling kernel 1.48 -> 1.35 ms); the fp32-MFMA
# kernels (launch_f32.hip) are slower that way and keep the compiler's default (AGPR accumulators).
TU_FLAGS = {"adanerf_hip.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def source_hash():
    """sha1 over the library's sources (csrc + the ABI header): stamps profiles so that bench.py can tell a PMC summary
    taken with other kernels from a current one."""
    import hashlib
    h = hashlib.sha1()
    for d in sorted(LIB_DEPS):
        path = os.path.join(CSRC, d)
        if os.path.exists(path) and not os.path.basename(d).startswith("x_"):     # x_*: experiment-only headers, not in the shipped library
            h.update(d.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def library_path():
    return os.path.join(LIBDIR, "libadanerf_hip.so")


def cli_path():
    return os.path.join(BINDIR, "adanerf")


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X build needs the ROCm toolchain")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, out=None, extra_flags=(), tu_flags=None):
    """hipcc --offload-arch=gfx950: one object per translation unit (TU_FLAGS), then -shared ->
    adanerf_amd/lib/libadanerf_hip.so (cross-compiles without a GPU).  ``out`` / ``extra_flags`` / ``tu_flags`` ({source: [flags]},
    on top of TU_FLAGS): experiment variants (tools/ablate.sh)."""
    variant = out is not None
    out = out or library_path()
    deps = [os.path.join(CSRC, d) for d in LIB_DEPS]
    if not variant and os.environ.get("ADANERF_LIB"):
        return os.environ["ADANERF_LIB"]      # pre-built variant selected by the caller
    if not force and not variant and not _stale(out, deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj" if not variant else "obj_" + os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in LIB_SOURCES:
        obj = os.path.join(objdir, src + ".o")
        exp = ["-I", os.path.join(os.path.dirname(HERE), "tools", "experiments")] if "-DADN_EXPERIMENT" in extra_flags else []
        cmd = [_hipcc()] + HIPCC_FLAGS + TU_FLAGS.get(src, []) + exp + list(extra_flags) + list((tu_flags or {}).get(src, [])) + ["-fPIC", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-fPIC", "-shared"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


def build_cli(force=False, verbose=False):
    """Headless `adanerf` host executable (same CLI as the reference viewer) linked against the library."""
    lib = build_library(force=force, verbose=verbose)
    out = cli_path()
    srcs = [os.path.join(HOST, f) for f in sorted(os.listdir(HOST)) if f.endswith(".cpp")] if os.path.isdir(HOST) else []
    if not srcs:
        return None
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")] + [lib]
    if not force and not _stale(out, deps):
        return out
    os.makedirs(BINDIR, exist_ok=True)
    cxx = shutil.which("g++") or shutil.which("c++") or _hipcc()     # pure host C++ over the C ABI
    cmd = [cxx, "-O2", "-std=c++17"] + srcs + ["-L", LIBDIR, "-ladanerf_hip", "-Wl,-rpath,$ORIGIN/../lib", "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


def build_probes(force=False, verbose=False):
    """Device-function test helpers (tools/probes/*.hip -> adanerf_amd/bin/<name>); they include the kernel header,
    so the GPU tests can exercise device functions that have no ABI entry of their own (sin_or_cos)."""
    pdir = os.path.join(os.path.dirname(HERE), "tools", "probes")
    outs = []
    for name in ("sincos_probe", "mfma_peak"):      # mfma_peak: the table behind bench.py's roofline.sustained_peak (csrc/k_probe.hip.hpp)
        src = os.path.join(pdir, name + ".hip")
        out = os.path.join(BINDIR, name)
        if not os.path.exists(src):
            continue
        if force or _stale(out, [src] + [os.path.join(CSRC, d) for d in LIB_DEPS]):
            os.makedirs(BINDIR, exist_ok=True)
            cmd = [_hipcc()] + HIPCC_FLAGS + ["-I", CSRC, src, "-o", out]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        outs.append(out)
    return outs


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--out", default=None, help="build an experiment variant of the library to this path instead")
    ap.add_argument("--flags", default="", help="extra hipcc flags for every translation unit (quoted string)")
    ap.add_argument("--tu-flags", action="append", default=[], metavar="SOURCE=FLAGS",
                    help="extra hipcc flags for ONE translation unit, e.g. --tu-flags 'launch_f32.hip=-fno-slp-vectorize' (repeatable)")
    a = ap.parse_args()
    tu = {}
    for spec in a.tu_flags:
        src, _, fl = spec.partition("=")
        if src not in LIB_SOURCES:
            raise SystemExit("--tu-flags: %s is not one of %s" % (src, LIB_SOURCES))
        tu.setdefault(src, []).extend(fl.split())
    if a.out:
        print(build_library(force=True, verbose=True, out=os.path.abspath(a.out), extra_flags=a.flags.split(), tu_flags=tu))
    else:
        print(build_library(force=True, verbose=True, extra_flags=a.flags.split(), tu_flags=tu))
        print(build_cli(force=True, verbose=True))
        print(build_probes(force=True, verbose=True))
