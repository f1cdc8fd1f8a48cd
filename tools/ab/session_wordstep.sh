#!/bin/bash
# A/B: element rows finished like operand rows - one Montgomery step in the word domain (table carries 2^32), then the re-cut into limbs
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
{
WORKLOADS="c2 c3 k3 w5 w7 h3 h9" STEPS=20 bash tools/ab/ab.sh
WORKLOADS="c5" BENCH_ARGS="--total-log2 21" STEPS=30 bash tools/ab/ab.sh
} > gpurun_out/ab_word_step.txt 2>&1
cat gpurun_out/ab_word_step.txt
