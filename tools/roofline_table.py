#!/usr/bin/env python3
"""The table of DESIGN.md section 3.5 from committed evidence: profiles/r06/<prefix>bench_*.json (bench lines), profiles/hbm_traffic.json
(FETCH_SIZE / WRITE_SIZE passes) and profiles/valu_instructions.json (SQ_INSTS_VALU passes).  usage: tools/roofline_table.py [prefix = z_]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else "z_"
tr = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
vi = json.load(open(os.path.join(ROOT, "profiles", "valu_instructions.json")))
LINES = [("c2", "**c2**: BLS12-381 Fr t = 3, 8 + 31, 2^20 states (BASELINE configs[1])"),
         ("c2_2e21", "the same, 2^21 states (per-GPU shard of configs[3])"),
         ("k3", "**k3**: t = 3, α = 17 (the reference's default rate-2 parameters), 2^20 states"),
         ("c3", "**c3**: BN254 Fr t = 9, 8 + 57, 2^18 states (configs[2])"),
         ("c5_2e21", "**c5**, 2^21 leaves (per-GPU subtree of configs[4]), 21 launches"),
         ("c5", "**c5**, 2^24 leaves on one GPU, 24 launches"),
         ("h3", "**h3**: `absorb(4); squeeze(1)` per row, t = 3, 2^20 rows (2 permutations each)"),
         ("h9", "**h9**: `absorb(8); squeeze(1)`, t = 9, 2^18 rows"),
         ("d3", "**d3**: duplex driver, `absorb(4)` + `squeeze(3)` per step, 2^20 sponges (4 permutations each)"),
         ("d9", "**d9**: `absorb(11)` + `squeeze(9)`, t = 9, 2^18 sponges")]
LINES += [("w%d" % t, "**w%d**: BLS12-381 Fr t = %d, 8 + %d, 2^%d states" % (t, t, 56 if t < 6 else 57, 19 if t < 6 else 18)) for t in (4, 5, 6, 7, 8)]
print("| workload (`bench.py --workload`) | engine (`HE` = `HybridEngine`) | permutations/s (at the box's clock) | at 2.2 GHz | kernel time per step | algorithmic HBM rate = fraction of 8 TB/s "
      "| HBM traffic (PMC) / algorithmic | VALU instructions per permutation | clocks per VALU instruction and SIMD (4.0 = every slot) | `valu_issue.frac` |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, label in LINES:
    path = os.path.join(ROOT, "profiles", "r06", "%sbench_%s.json" % (prefix, name))
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    t, v = tr.get(name), vi.get(name)
    t = t if isinstance(t, dict) else None
    algo = d["roofline"]["algorithmic_bytes_per_launch"]
    traffic = "%.1f / %.1f MB = %.2f×" % (t["bytes_per_launch"] / 1e6, algo / 1e6, t["bytes_per_launch"] / algo) if t else "not measured"
    valu = "%d" % v["valu_instructions_per_permutation"] if v else "see `%svalu_driver_and_tree_kernels.txt`" % prefix
    frac = (d.get("valu_issue") or {}).get("frac")
    cn = d.get("clock_normalised") or {}
    clk = cn.get("shader_clock_hz") or d["int_valu"]["shader_clock_hz"]
    cpi = cn.get("clocks_per_valu_instruction_and_simd")
    print("| %s | %s | %.3g (%.2f GHz) | %.3g | %.3f ms | %.1f GB/s = %.4f | %s | %s | %s | %s |" % (
        label, d["engine"]["name"].replace("HybridEngine", "HE"), d["value"], clk / 1e9, d["value"] * 2.2e9 / clk, d["roofline"]["kernel_ms"], d["roofline"]["achieved"],
        d["roofline"]["frac"], traffic, valu, "%.2f" % cpi if cpi else "-", "%.2f" % frac if frac else "-"))
