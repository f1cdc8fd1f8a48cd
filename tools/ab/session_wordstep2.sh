#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
{
WORKLOADS="c5" STEPS=4 bash tools/ab/ab.sh
WORKLOADS="c5" BENCH_ARGS="--total-log2 21" STEPS=30 bash tools/ab/ab.sh
} > gpurun_out/ab_word_step_c5.txt 2>&1
cat gpurun_out/ab_word_step_c5.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/word_step_gpu_suite.txt 2>&1
grep -E "passed|failed" gpurun_out/word_step_gpu_suite.txt
