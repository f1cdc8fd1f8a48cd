#!/bin/bash
# A/B of the tree-level switch points in one session: per-level times of a 2^21-leaf tree for each variant library.
# usage (GPU box, repo root): bash tools/ab/session_levels.sh "<variants>" <out file>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
VARS=${1:-"base"}
OUT=${2:-gpurun_out/ab_levels.txt}
mkdir -p $(dirname $OUT)
cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
for round in 1 2; do
  for v in $VARS; do
    cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so
    python tools/merkle_levels.py 21 2>/dev/null | grep "^2\^" | awk -v v=$v -v r=$round '$1 ~ /2\^(1[5-9]|2[01])/ {print v, "round", r, $0}'
  done
done | tee $OUT
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
