cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
( timeout 1800 python -m pytest tests -x -q -m gpu ) > gpurun_out/r03g/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03g/pytest_gpu.log | tail -3
WORKLOADS="c2 c3 h3 w4 w5 w6 w7 w8" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee gpurun_out/r03g/ab_norm.txt
WORKLOADS="c5" STEPS=10 BENCH_ARGS="--total-log2 21" bash tools/ab/ab.sh 2>&1 | tee -a gpurun_out/r03g/ab_norm.txt
python tools/merkle_levels.py 16 2>/dev/null | head -16 | tee gpurun_out/r03g/levels.txt
