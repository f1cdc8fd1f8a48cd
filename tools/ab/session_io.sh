cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03l
( timeout 1800 python -m pytest tests -x -q -m gpu ) > gpurun_out/r03l/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03l/pytest_gpu.log | tail -3
WORKLOADS="c2 c3 h3 h9" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee gpurun_out/r03l/ab_io.txt
WORKLOADS="c5" STEPS=10 BENCH_ARGS="--total-log2 21" bash tools/ab/ab.sh 2>&1 | tee -a gpurun_out/r03l/ab_io.txt
