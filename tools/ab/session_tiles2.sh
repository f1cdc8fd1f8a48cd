#!/bin/bash
# A/B: two LDS tiles (one workgroup barrier per row instead of two) at t = 3, 4, 6, 7
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
cp sponge_amd/libposeidon_mi355x.so /tmp/keep.so
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/tiles2_parity.txt 2>&1
grep -E "passed|failed" gpurun_out/tiles2_parity.txt
cp /tmp/keep.so sponge_amd/libposeidon_mi355x.so
{
WORKLOADS="c2 k3 h3 w4 w6 w7 d3" STEPS=20 bash tools/ab/ab.sh
WORKLOADS="c5" STEPS=4 bash tools/ab/ab.sh
} > gpurun_out/ab_tiles2.txt 2>&1
cat gpurun_out/ab_tiles2.txt
