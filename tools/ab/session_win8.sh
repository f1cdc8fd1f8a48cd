#!/bin/bash
# A/B: windows of 8 (new) against 7 (old) at t = 8, 9
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
{
WORKLOADS="c3 h9 d9 w8" STEPS=20 bash tools/ab/ab.sh
} > gpurun_out/ab_win8.txt 2>&1
cat gpurun_out/ab_win8.txt
