#!/usr/bin/env python
"""Heuristic ISA scan for the hazard of DESIGN.md section 3.1d: a >= 128-bit VMEM store whose data registers are written
again within the next few instructions of straight-line code.  usage: store_war_scan.py file.s [window=6]"""
import re
import sys

win = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lines = [l.rstrip() for l in open(sys.argv[1])]
kernel = "?"
hits = {}
ins = []                                       # (kernel, text)
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kernel = m.group(1)
    t = l.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        if t.endswith(":") and not t.startswith(";"):
            ins.append((kernel, "LABEL"))
        continue
    ins.append((kernel, t.split(";")[0].strip()))


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


for i, (k, t) in enumerate(ins):
    m = re.match(r"(global_store_dwordx[34]|buffer_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)", t)
    if not m:
        continue
    ops = [o.strip() for o in m.group(2).split(",")]
    data = regs(ops[1]) if m.group(1).startswith(("global", "flat", "scratch")) else regs(ops[0])
    for d in range(1, win + 1):
        if i + d >= len(ins) or ins[i + d][1] == "LABEL" or ins[i + d][1].startswith(("s_branch", "s_cbranch", "s_endpgm")):
            break
        nt = ins[i + d][1]
        mm = re.match(r"(v_\w+|ds_read\w*|ds_bpermute\w*|global_load\w*|buffer_load\w*|v_mfma\w*)\s+([^,]+)", nt)
        if mm and regs(mm.group(2).strip()) & data and not nt.startswith(("v_cmp", "v_cmpx")):
            hits.setdefault(k, []).append((d, t[:60], nt[:70]))
            break
for k, v in hits.items():
    import subprocess
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:90]
    print("%s: %d stores whose data registers are rewritten within %d instructions; nearest: %d" % (name, len(v), win, min(x[0] for x in v)))
    for d, a, b in sorted(v)[:2]:
        print("      +%d  %s   ->   %s" % (d, a, b))
