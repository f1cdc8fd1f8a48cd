#!/bin/bash
# A/B of builds of the wide-state absorb / squeeze driver in one session (same device), interleaved rounds.
# usage (GPU box, repo root): bash tools/ab/session_driver.sh "<variants, e.g. r3drv head>" <out file> [rounds]
# (tools/ab/libposeidon_<variant>.so: tools/ab/build_variant.sh; r3drv = -DPMX_HYB_PASS_MIN_T=10, the per-lane-loop kernels of round 3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
VARS=${1:-"r3drv head"}
OUT=${2:-gpurun_out/ab_driver.txt}
ROUNDS=${3:-2}
mkdir -p $(dirname $OUT)
cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
WIDE="--field bn254_fr --rate 8 --rounds 8 57 --log2 18 --absorb 11 --squeeze 9"
for round in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so
    ( python tools/sponge_rate.py $WIDE; python tools/sponge_rate.py $WIDE --mixed
      python tools/sponge_rate.py --rate 7 --log2 18 --absorb 10 --squeeze 8; python tools/sponge_rate.py --rate 7 --log2 18 --absorb 10 --squeeze 8 --mixed
      python tools/sponge_rate.py --rate 6 --log2 18 --absorb 9 --squeeze 7 --mixed
      python tools/sponge_rate.py --rate 5 --log2 18 --absorb 8 --squeeze 6 --mixed
      python tools/sponge_rate.py --rate 4 --log2 19 --absorb 7 --squeeze 5 --mixed
      python tools/sponge_rate.py --rate 3 --log2 19 --absorb 6 --squeeze 4 --mixed ) 2>/dev/null | sed "s/^/$v round $round: /"
  done
done | tee $OUT
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
