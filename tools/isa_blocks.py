#!/usr/bin/env python3
"""Instruction histogram per basic block of one kernel in sponge_amd/csrc/build/pmx_device.s (make -C sponge_amd/csrc asm TU=n).
usage: isa_blocks.py <substring of the mangled kernel name> [min block size = 200]"""
import collections
import re
import sys

s = open("sponge_amd/csrc/build/pmx_device.s").read()
want = sys.argv[1]
min_size = int(sys.argv[2]) if len(sys.argv) > 2 else 200
m = re.search(r"^(\S*%s\S*):" % re.escape(want), s, re.M)
name = m.group(1)
i = m.start()
j = s.index(".Lfunc_end", i)
blocks, cur = [], ("entry", [])
for l in s[i:j].split("\n"):
    l = l.strip()
    if re.match(r"^\.?LBB\d+_\d+:", l):
        blocks.append(cur)
        cur = (l[:-1], [])
        continue
    if not l or l.startswith(";") or l.startswith("."):
        continue
    cur[1].append(l.split()[0])
blocks.append(cur)
print(name)
tot = collections.Counter()
for n, ins in blocks:
    c = collections.Counter(ins)
    tot.update(c)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    mad = c.get("v_mad_u64_u32", 0)
    if len(ins) >= min_size:
        print("%-10s %6d instr  valu %6d  mad %6d (%.3f)  s_load %4d  waitcnt %4d  ds %4d  lane-moves %4d  scratch %3d" % (
            n, len(ins), valu, mad, mad / max(valu, 1), sum(v for k, v in c.items() if k.startswith("s_load")), c.get("s_waitcnt", 0),
            sum(v for k, v in c.items() if k.startswith("ds_")), c.get("v_readlane_b32", 0) + c.get("v_writelane_b32", 0),
            sum(v for k, v in c.items() if k.startswith("scratch_"))))
        print("      ", [(k, v) for k, v in c.most_common(16) if k != "v_mad_u64_u32"])
valu = sum(v for k, v in tot.items() if k.startswith("v_"))
print("total %d instructions, %d VALU, %d mads" % (sum(tot.values()), valu, tot.get("v_mad_u64_u32", 0)))
