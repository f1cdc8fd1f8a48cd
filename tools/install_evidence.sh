#!/bin/bash
# Copies the summaries of gpurun_out/<tags...> (tools/gpu_r06.sh sessions of ONE build) into profiles/r06/<prefix>_*: bench lines, kernel stats,
# counter summaries, test tails.  Scratch stays in gpurun_out/.   usage: tools/install_evidence.sh <prefix> <tag> [<tag> ...]
set -e
P=$1; shift
R=$(cd $(dirname $0)/.. && pwd); D=$R/profiles/r06
strip() { grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname \|^Librccl path\|amdgpu.ids" "$1" > "$2" || true; }
for TAG in "$@"; do
  S=$R/gpurun_out/$TAG
  for f in $S/bench_*.json; do [ -s "$f" ] && cp $f $D/${P}_$(basename $f); done
  for f in $S/*_kernel_stats.csv; do [ -s "$f" ] && cp $f $D/${P}_$(basename $f); done
  [ -f $S/device.txt ] && cp $S/device.txt $D/${P}_device.txt
  [ -f $S/pytest_gpu.log ] && { strip $S/pytest_gpu.log /tmp/_a; { echo "== shipped library"; tail -22 /tmp/_a; } > $D/${P}_pytest_gpu_tail.txt; }
  [ -f $S/pytest_gpu_test_library.log ] && { strip $S/pytest_gpu_test_library.log /tmp/_a; { echo "== --pmx-test-library"; tail -6 /tmp/_a; } >> $D/${P}_pytest_gpu_tail.txt; }
  [ -f $S/smoke.log ] && strip $S/smoke.log $D/${P}_smoke.txt
  for n in fuzz_configs sponge_rate merkle_levels host_path valu_driver_and_tree_kernels pmc_c3_stalls pmc_c2_stalls; do [ -f $S/$n.txt ] && strip $S/$n.txt $D/${P}_$n.txt; done
  [ -f $S/valu_count.log ] && strip $S/valu_count.log $D/${P}_valu_count.txt
  [ -f $S/traffic.log ] && strip $S/traffic.log $D/${P}_hbm_traffic.txt
  if [ -d $S/group_rehearsal ]; then rm -rf $D/${P}_group_rehearsal; mkdir -p $D/${P}_group_rehearsal; cp $S/group_rehearsal/*.json $S/group_rehearsal/*.txt $D/${P}_group_rehearsal/ 2>/dev/null || true; [ -f $S/group_rehearsal.txt ] && strip $S/group_rehearsal.txt $D/${P}_group_rehearsal/summary.txt; fi
done
# (profiles/r06/<prefix>_pmc_fetch_smem_lds_levels_c3_c2.txt is made by hand from tools/pmc_fetch_levels.sh: see profiles/r05/README.md)
# the counter JSONs bench.py quotes: say where they were taken
python3 - "$P" <<PY
import json, sys
p = "$R/profiles/hbm_traffic.json"
d = json.load(open(p))
for k in d:
    if isinstance(d[k], dict):
        d[k].setdefault("taken", "round 6 (profiles/r06/%s_*), profiles/r06/%s_hbm_traffic.txt" % (sys.argv[1], sys.argv[1]))
json.dump(d, open(p, "w"), indent=1)
PY
ls $D | grep "^${P}_" | wc -l
