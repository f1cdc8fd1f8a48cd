#!/usr/bin/env python3
"""Run-length view of the instruction stream around the matrix-core products of a kernel's ISA (hipcc -S): which classes of
instructions sit between consecutive v_mfma - i.e. whether a row's finish is issued BESIDE the next row's products or after them.
usage: tools/isa_mfma_runs.py <file.s> [first_line last_line]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, len(lines))
def cls(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("v_permlane"): return "swap"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_barrier"): return "BARRIER"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("scratch_"): return "SCRATCH"
    if op.startswith("global_") or op.startswith("buffer_"): return "vmem"
    if op.startswith("s_load"): return "smem"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return None
prev, n, start = None, 0, 0
out = []
for i in range(lo, min(hi, len(lines))):
    l = lines[i].strip()
    if not l or l.startswith(";") or l.startswith("."):
        if l.startswith(".LBB"):
            if prev: out.append("%s x%d" % (prev, n))
            out.append("\n%d %s" % (i + 1, l.split()[0])); prev, n = None, 0
        continue
    c = cls(l.split()[0])
    if c is None: continue
    if c == prev: n += 1
    else:
        if prev: out.append("%s x%d" % (prev, n))
        prev, n = c, 1
if prev: out.append("%s x%d" % (prev, n))
print(" ".join(out))
