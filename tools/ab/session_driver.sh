#!/bin/bash
# A/B of the wide-state absorb / squeeze driver forms in one session (same device), interleaved rounds.
# usage (GPU box, repo root): bash tools/ab/session_driver.sh "<variants, e.g. v1 v3 v4>" <out file>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
VARS=${1:-"v1 v3"}
OUT=${2:-gpurun_out/ab_driver.txt}
mkdir -p $(dirname $OUT)
cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
WIDE="--field bn254_fr --rate 8 --rounds 8 57 --log2 18 --absorb 11 --squeeze 9"
for round in 1 2 3; do
  for v in $VARS; do
    cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so
    ( python tools/sponge_rate.py $WIDE; python tools/sponge_rate.py $WIDE --mixed
      python tools/sponge_rate.py --rate 6 --log2 18 --absorb 9 --squeeze 7 --mixed
      python tools/sponge_rate.py --rate 3 --log2 19 --absorb 6 --squeeze 4 --mixed ) 2>/dev/null | sed "s/^/$v round $round: /"
  done
done | tee $OUT
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
