#!/usr/bin/env python3
"""Median per-launch value of every counter the pmc_* passes of tools/gpu_r02.sh collected, for the dominant kernel.
usage: pmc_summary.py <out_dir>"""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "pmc_[a-z]_c*"))):
    if not os.path.isdir(d):
        continue
    vals = collections.defaultdict(list)
    kern = None
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if "permute_kernel" in row["Kernel_Name"]:
                kern = row["Kernel_Name"].split("(")[0]
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(os.path.basename(d), kern)
    for k in sorted(vals):
        v = sorted(vals[k])
        print("   %-32s %16.0f   (%d launches)" % (k, v[len(v) // 2], len(v)))
