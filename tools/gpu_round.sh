#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, microbenchmark, rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/device.txt
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
timeout 600 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 600 python bench.py --workload c3 --steps 5 --warmup 1 --cpu-seconds 6 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
timeout 300 tools/valu_microbench > $OUT/valu_microbench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python3 $R/bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_c3.log 2>&1
cd $R
tail -3 $OUT/pytest_gpu.log; cat $OUT/smoke.log | tail -2; cat $OUT/bench_c2.json; cat $OUT/bench_c3.json; tail -45 $OUT/valu_microbench.txt
