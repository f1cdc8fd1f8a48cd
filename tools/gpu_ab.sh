#!/bin/bash
# quick A/B probe: a parity subset, then bench lines.  Usage: bash tools/gpu_ab.sh <tag> "<workloads>" [pytest -k expr]
TAG=${1:-ab}
WL=${2:-"c2 c3 c5s h9"}
KEXPR=${3:-"golden or ragged or widths or run_time"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$KEXPR" ) > $OUT/pytest_gpu.log 2>&1
tail -2 $OUT/pytest_gpu.log
for w in $WL; do
  case $w in
    c5s) args="--workload c5 --total-log2 21 --steps 10 --warmup 2";;
    c5) args="--workload c5 --steps 3 --warmup 1";;
    c2) args="--workload c2 --steps 20 --warmup 5";;
    *) args="--workload $w --steps 8 --warmup 2";;
  esac
  for rep in 1 2; do
    timeout 600 python bench.py $args --no-cpu-baseline > $OUT/bench_${w}_$rep.json 2> $OUT/bench_${w}_$rep.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${w}_$rep.json"))
    print("$w", "$rep", "%.4g perm/s"%d["value"], "ms/step %.3f"%d["ms_per_step"], "mad frac %.3f"%d["int_valu"]["frac"], "peak %.3g"%d["int_valu"]["peak"], "verified", d["verified"])
except Exception as e:
    print("$w unreadable", e); print(open("$OUT/bench_${w}_$rep.err").read()[-800:])
PY
  done
done
