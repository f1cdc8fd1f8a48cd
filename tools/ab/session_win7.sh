#!/bin/bash
# A/B: windows of 7 S-boxes at t = 7, 8, 9 (6 since round 4, when the history terms were VALU products)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
{
WORKLOADS="c3 w7 w8 h9 d9" STEPS=20 bash tools/ab/ab.sh
} > gpurun_out/ab_win7.txt 2>&1
cat gpurun_out/ab_win7.txt
