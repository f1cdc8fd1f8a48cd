#!/usr/bin/env python3
"""VALU issue-slot accounting of one kernel from a rocprofv3 --pmc pass (tools/gpu_r03.sh, stage `slots`).

usage: valu_slots.py <counter dir> <kernel substring> <units per launch> <lanes per unit> <mads per unit> [<avg kernel ns>]

Prints the medians of the SQ counters over the launches of the kernel, the VALU instructions per unit (one unit = one
permutation; SQ_INSTS_VALU counts wave instructions, every lane of a wave executes all of them), the share of them that
are multiplies, and - given the kernel's average duration from the kernel-trace pass - the time one SIMD spends per VALU
instruction (tools/issue_model_microbench.hip: a SIMD takes one VALU instruction per ~4 shader cycles whenever the stream
holds multiplies, whichever wave it comes from)."""
import collections
import csv
import glob
import os
import sys

d, sub, units, lanes, mads = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
avg_ns = float(sys.argv[6]) if len(sys.argv) > 6 else None
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        if sub in name:
            vals[name.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in vals.items():
    med = {k: sorted(v)[len(v) // 2] for k, v in cs.items()}
    print(name[:120], "(%d launches)" % len(next(iter(cs.values()))))
    for k in sorted(med):
        print("    %-24s %18.0f" % (k, med[k]))
    if "SQ_INSTS_VALU" in med:
        waves = med.get("SQ_WAVES") or units * lanes / 64.0
        per_lane = med["SQ_INSTS_VALU"] / waves   # what every lane of a wave executes = per unit when one lane holds one unit
        print("    VALU instructions per wave %.0f over %.0f waves; multiplies per unit %.0f = %.1f %% of them" % (per_lane, waves, mads / lanes, 100.0 * mads / lanes / per_lane))
        if avg_ns:
            print("    one SIMD: %.3f ns per VALU instruction over the kernel's %.1f us" % (avg_ns * 256 * 4 / med["SQ_INSTS_VALU"], avg_ns / 1e3))
    if med.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_VALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
            if k in med:
                print("    %-24s / SQ_WAVE_CYCLES %6.1f %%" % (k, 100 * med[k] / med["SQ_WAVE_CYCLES"]))
    if med.get("SQ_BUSY_CYCLES") and "SQ_INSTS_VALU" in med:
        print("    SQ_INSTS_VALU / SQ_BUSY_CYCLES %.3f" % (med["SQ_INSTS_VALU"] / med["SQ_BUSY_CYCLES"]))
