#!/bin/bash
# A/B: the window of the partial rounds as long as the width (t = 7: 7, 8: 8, 9: 9; history tables not fetched ahead at t = 9) against 6
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
{
WORKLOADS="c3 w7 w8 h9 d9 c2" STEPS=20 bash tools/ab/ab.sh
} > gpurun_out/ab_window_width.txt 2>&1
cat gpurun_out/ab_window_width.txt
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" ; timeout 900 python tools/diag/fuzz_configs.py 300 909 2>&1 | tail -1 ) > gpurun_out/window_width_suite.txt 2>&1
cat gpurun_out/window_width_suite.txt
