#!/usr/bin/env python3
"""profiles/valu_instructions.json entry of one workload from a rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES pass of its bench
command: VALU instructions per wave (= per permutation: one state per lane) of the dominant kernel.
usage: valu_count.py <pmc dir> <kernel substring> <workload> <permutations per launch> <engine name> <waves per simd> <out.json> [source] [permutations per lane]"""
import collections
import csv
import glob
import json
import os
import sys

d, sub, workload, units, engine, waves, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], int(sys.argv[6]), sys.argv[7]
source = sys.argv[8] if len(sys.argv) > 8 else ""
per_lane = int(sys.argv[9]) if len(sys.argv) > 9 else 1
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if sub in row["Kernel_Name"]:
            vals[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
if not vals:
    raise SystemExit("no kernel matching %r in %s" % (sub, d))
name, cs = max(vals.items(), key=lambda kv: sorted(kv[1]["SQ_INSTS_VALU"])[len(kv[1]["SQ_INSTS_VALU"]) // 2])
med = {k: sorted(v)[len(v) // 2] for k, v in cs.items()}
rec = {"kernel": name.replace("void ", ""), "engine": engine, "units_per_launch": units, "waves_per_simd": waves,
       "SQ_INSTS_VALU_median": med["SQ_INSTS_VALU"], "SQ_WAVES": med["SQ_WAVES"],
       "valu_instructions_per_wave": round(med["SQ_INSTS_VALU"] / med["SQ_WAVES"]), "permutations_per_lane": per_lane,
       "valu_instructions_per_permutation": round(med["SQ_INSTS_VALU"] / med["SQ_WAVES"] / per_lane), "launches_seen": len(cs["SQ_INSTS_VALU"]), "source": source}
data = json.load(open(out)) if os.path.exists(out) else {}
data.pop("floor_ns_per_instruction", None)      # the floor is measured in the run now (pmx_diag_issue_slot)
data.pop("floor_source", None)
data[workload] = rec
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hash import kernel_source_hash
data["_kernel_source_hash"] = kernel_source_hash()     # the device code these figures were taken on (tools/source_hash.py)
json.dump(data, open(out, "w"), indent=1)
print(json.dumps(rec))
