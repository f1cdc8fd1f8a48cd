#!/bin/bash
# the full rounds' S-boxes two per trip (-DPMX_SBOX_PAIRS) against HEAD
cd ${GRAFT_REPO_ROOT:-/root/repo}
VARIANTS="head pairs" WORKLOADS="c2 c3 w6 w8 h9 k3 w4" ROUNDS=3 bash tools/ab/ab_multi.sh
