#!/bin/bash
# A/B: a window's free scale chosen so that its history constant is 1 .. 4 (t = 3, alpha = 5: lazy additions instead of a table product)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
{
WORKLOADS="c2 h3 d3" STEPS=20 bash tools/ab/ab.sh
WORKLOADS="c5" STEPS=4 bash tools/ab/ab.sh
WORKLOADS="c5" BENCH_ARGS="--total-log2 21" STEPS=10 bash tools/ab/ab.sh
} > gpurun_out/ab_small_hist.txt 2>&1
cat gpurun_out/ab_small_hist.txt
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" ) > gpurun_out/small_hist_suite.txt 2>&1
cat gpurun_out/small_hist_suite.txt
