#!/usr/bin/env python3
"""Register / scratch summary of a -Rpass-analysis=kernel-resource-usage report (hipcc ... 2> report): one line per kernel.
usage: tools/rpt_summary.py <report> [substring filter]"""
import re, subprocess, sys
rpt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r'remark: Function Name: ', rpt)[1:]:
    mangled = b.split()[0]
    name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name.replace('pmx::', '').replace('void ', ''))
    if flt and flt not in name:
        continue
    g = lambda k: int(re.search(k + r': (\d+)', b).group(1))
    print("%-70s VGPR %3d AGPR %3d scratch %4d occ %d sgpr-spill %3d" % (name[:70], g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g('SGPRs Spill')))
