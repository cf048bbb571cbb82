#!/usr/bin/env python3
"""Static check of the hand-issued LDS reads (HsLayer, k_mlp16.hip.hpp): walks the gfx950 assembly of the kernels matching
a regex in program order with a model of the in-order LDS return queue (ds_read / ds_write push, `s_waitcnt lgkmcnt(N)`
retires all but the N youngest) and reports every instruction that touches a VGPR an LDS read has not delivered yet -- a
too-large hand-counted wait, or a copy the compiler placed between an asm block that issues a read and the one that waits.
Run after tools/asm_report.py (which writes /tmp/adanerf_all.s)."""
import re, sys
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else "mlp16")
src = open("/tmp/adanerf_all.s").read()
reg = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")      # architectural and accumulation VGPRs
def regs(text):
    out = set()
    for m in reg.finditer(text):
        if m.group(1): out.add((m.group(1), int(m.group(2))))
        else: out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out
bad = 0
for f in re.split(r"\n\s*\.globl\s+", src):
    name = f.split("\n", 1)[0].strip()
    if not pat.search(name) or "v_mfma" not in f:
        continue
    body = f.split(".amdhsa_kernel")[0] if ".amdhsa_kernel" in f else f
    queue = []          # (dest regs, line no) of LDS operations in flight, oldest first
    n_reads = n_checked = 0
    valu_age = {}       # register -> wait states since a (non-MFMA) VALU instruction wrote it
    n_valu_hazard = 0
    for ln, raw in enumerate(body.split("\n")):
        l = raw.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        op = l.split()[0]
        # second check: the compiler keeps 2 wait states between a VALU write and an MFMA that reads the register, but it
        # cannot see the MFMAs inside asm blocks
        states = (int(l.split()[1]) + 1) if op == "s_nop" else 1
        if op.startswith("v_mfma"):
            srcs = regs(",".join(l[len(op):].split(",")[1:]))
            close = sorted(r for r in srcs if valu_age.get(r, 99) < 2)
            if close:
                n_valu_hazard += 1; bad += 1
                print("%s: line %d `%s` reads %s %d wait state(s) after a VALU write" % (name[:60], ln, l, close[:4], valu_age[close[0]]))
        for r in list(valu_age):
            valu_age[r] += states
            if valu_age[r] > 8: del valu_age[r]
        if op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_cmp"):
            for r in regs(l[len(op):].split(",")[0]): valu_age[r] = 0
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                while len(queue) > n: queue.pop(0)
            elif "lgkmcnt" not in l and re.search(r"s_waitcnt\s+(0x[0-9a-f]+|\d+)$", l):
                queue.clear()       # raw immediate: treat as a full wait
            continue
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            queue.append((set(), ln)); continue        # scalar loads share the counter (no VGPR destination)
        operands = l[len(op):]
        touched = regs(operands)
        pending = set().union(*[q[0] for q in queue]) if queue else set()
        if op.startswith("ds_read") or op.startswith("ds_write"):
            first = operands.split(",")[0]
            dest = regs(first) if op.startswith("ds_read") else set()
            addr_and_data = regs(",".join(operands.split(",")[1:])) if op.startswith("ds_read") else touched
            hit = (addr_and_data | dest) & pending
            if hit:
                bad += 1; print("%s: line %d `%s` touches in-flight v%s" % (name[:60], ln, l, sorted(hit)[:4]))
            queue.append((dest, ln)); n_reads += 1
            continue
        hit = touched & pending
        n_checked += 1
        if hit:
            bad += 1
            print("%s: line %d `%s` touches in-flight v%s" % (name[:60], ln, l, sorted(hit)[:4]))
    print("%s: %d LDS operations, %d other instructions checked" % (name[:70], n_reads, n_checked))
print("in-flight violations:", bad)
sys.exit(1 if bad else 0)
