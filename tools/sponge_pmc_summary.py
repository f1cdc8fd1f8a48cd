#!/usr/bin/env python3
"""Per-kernel summary of the absorb / squeeze driver profiles of tools/gpu_r03.sh (stage `sponge`): average duration from the
kernel-trace stats, VALU instructions per wave and the SQ cycle shares from the counter pass.  usage: sponge_pmc_summary.py <out_dir>"""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
for tag in ("sponge", "sponge_mixed"):
    print("==", tag)
    for path in glob.glob(os.path.join(out, "prof_" + tag, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if "absorb_kernel" in row["Name"] or "squeeze_kernel" in row["Name"]:
                print("  %-90s calls %5s  avg %10.1f us  min %10.1f us" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3))
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(out, "pmc_" + tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0]
            if "absorb_kernel" in name or "squeeze_kernel" in name:
                vals[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, cs in vals.items():
        med = {k: sorted(v)[len(v) // 2] for k, v in cs.items()}
        print("  ", name[:100])
        for k in sorted(med):
            print("      %-24s %16.0f" % (k, med[k]))
        if med.get("SQ_WAVES"):
            print("      VALU instructions per wave        %10.0f" % (med.get("SQ_INSTS_VALU", 0) / med["SQ_WAVES"]))
        if med.get("SQ_WAVE_CYCLES"):
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in med:
                    print("      %-24s / SQ_WAVE_CYCLES %6.1f %%" % (k, 100 * med[k] / med["SQ_WAVE_CYCLES"]))
