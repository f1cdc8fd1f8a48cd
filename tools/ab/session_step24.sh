#!/bin/bash
# A/B: the Montgomery step of a row inside its word sums (24 bits, one multiply-add per word) against a step of its own behind them (32 bits)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
{
WORKLOADS="c2 c3 k3 w5 w7 h3 h9" STEPS=20 bash tools/ab/ab.sh
WORKLOADS="c5" STEPS=4 bash tools/ab/ab.sh
WORKLOADS="c5" BENCH_ARGS="--total-log2 21" STEPS=10 bash tools/ab/ab.sh
} > gpurun_out/ab_step24.txt 2>&1
cat gpurun_out/ab_step24.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/step24_gpu_suite.txt 2>&1
grep -E "passed|failed" gpurun_out/step24_gpu_suite.txt
