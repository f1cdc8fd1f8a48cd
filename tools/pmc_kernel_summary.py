#!/usr/bin/env python3
"""Per-kernel medians of a rocprofv3 --pmc counter pass.  usage: pmc_kernel_summary.py <pmc_dir> [name-filter]"""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        if flt in name:
            vals[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in sorted(vals.items()):
    med = {k: sorted(v)[len(v) // 2] for k, v in cs.items()}
    print(name[:150], " launches:", max(len(v) for v in cs.values()))
    for k in sorted(med):
        print("      %-24s %16.0f" % (k, med[k]))
    if med.get("SQ_WAVES"):
        print("      VALU instructions per wave        %10.0f" % (med.get("SQ_INSTS_VALU", 0) / med["SQ_WAVES"]))
    if med.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in med:
                print("      %-24s / SQ_WAVE_CYCLES %6.1f %%" % (k, 100 * med[k] / med["SQ_WAVE_CYCLES"]))
