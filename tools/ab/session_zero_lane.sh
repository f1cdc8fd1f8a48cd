#!/bin/bash
# A/B: the round-0 S-box of a capacity lane known to be zero as a constant of the config (compress, first permutation of a hash row)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
{
WORKLOADS="c5 h3 h9" STEPS=10 bash tools/ab/ab.sh
WORKLOADS="c5" BENCH_ARGS="--total-log2 21" STEPS=30 bash tools/ab/ab.sh
WORKLOADS="c2" STEPS=20 bash tools/ab/ab.sh
} > gpurun_out/ab_zero_lane.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/zero_lane_gpu_suite.txt 2>&1
tail -3 gpurun_out/zero_lane_gpu_suite.txt
cat gpurun_out/ab_zero_lane.txt
