cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
( timeout 1800 python -m pytest tests -x -q -m gpu ) > gpurun_out/r03e/pytest_gpu.log 2>&1; tail -3 gpurun_out/r03e/pytest_gpu.log
WORKLOADS="c2 c3 h3 w4 w6" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee gpurun_out/r03e/ab_mu.txt
WORKLOADS="c5" STEPS=10 BENCH_ARGS="--total-log2 21" bash tools/ab/ab.sh 2>&1 | tee -a gpurun_out/r03e/ab_mu.txt
