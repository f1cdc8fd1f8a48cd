cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03t; mkdir -p $OUT
cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
for v in orig tm17 cq64; do
  if [ $v = orig ]; then cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so; else cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so; fi
  echo "== $v"; python tools/merkle_levels.py 21 2>/dev/null | grep "leaves" | tail -8
  python tools/single_latency.py 2>/dev/null | tail -8
done 2>&1 | tee $OUT/thresholds.txt
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
