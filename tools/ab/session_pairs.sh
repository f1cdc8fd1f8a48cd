#!/bin/bash
# A/B: the full rounds' S-boxes two at a time (independent chains side by side) at t >= 6
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
{
WORKLOADS="c3 w6 h9" STEPS=20 bash tools/ab/ab.sh
} > gpurun_out/ab_sbox_pairs.txt 2>&1
cat gpurun_out/ab_sbox_pairs.txt
