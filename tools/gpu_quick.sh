#!/bin/bash
# quick perf probe: bench lines only (after a parity smoke)
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or ragged" ) > $OUT/pytest_gpu.log 2>&1
for w in ${WORKLOADS:-c2 c3 c5 h3 h9}; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
tail -1 $OUT/smoke.log; tail -2 $OUT/pytest_gpu.log
for w in ${WORKLOADS:-c2 c3 c5 h3 h9}; do python - <<PY
import json
d=json.load(open("$OUT/bench_$w.json"))
print("$w", "%.4g perm/s"%d["value"], "ms/step %.3f"%d["ms_per_step"], "mad frac %.3f"%d["int_valu"]["frac"])
PY
done
