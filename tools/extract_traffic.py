#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into HBM bytes per launch of
the dominant kernel, with the gfx950 corrections of MI355X_MICROARCH.md section HBM: both counters are in KiB,
and FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced read stream, so it is doubled.
usage: extract_traffic.py <fetch_dir> <write_dir> <kernel-substring> <workload> <out.json> [launches_per_step] [units_per_step]
With launches_per_step (e.g. 21 tree levels) the counters of that many consecutive launches are summed into one step."""
import csv
import glob
import json
import os
import sys


def per_launch(directory, counter, kernel_sub):
    vals = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") == counter and kernel_sub in row.get("Kernel_Name", ""):
                vals.append(float(row["Counter_Value"]))
    return vals


def main():
    fetch_dir, write_dir, kernel_sub, workload, out = sys.argv[1:6]
    group = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    units = int(sys.argv[7]) if len(sys.argv) > 7 else None
    fetch = per_launch(fetch_dir, "FETCH_SIZE", kernel_sub)
    write = per_launch(write_dir, "WRITE_SIZE", kernel_sub)
    if not fetch or not write:
        raise SystemExit(f"no counter rows found (fetch {len(fetch)}, write {len(write)})")
    if group > 1:   # rows are in dispatch order: sum every `group` consecutive launches
        fetch = [sum(fetch[i:i + group]) for i in range(0, len(fetch) - group + 1, group)]
        write = [sum(write[i:i + group]) for i in range(0, len(write) - group + 1, group)]
    f_kib = sorted(fetch)[len(fetch) // 2]
    w_kib = sorted(write)[len(write) // 2]
    rec = {"kernel": kernel_sub, "launches_per_step": group, "steps_seen": [len(fetch), len(write)],
           "FETCH_SIZE_KiB_median": f_kib, "WRITE_SIZE_KiB_median": w_kib,
           "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 B per 128 B request)",
           "bytes_per_launch": (2 * f_kib + w_kib) * 1024}
    if units is not None:
        rec["units_per_launch"] = units     # permutations one step processes: bench.py quotes the figure only at this size
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    data[workload] = rec
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from source_hash import kernel_source_hash
    data["_kernel_source_hash"] = kernel_source_hash()     # the device code these figures were taken on (tools/source_hash.py)
    json.dump(data, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
