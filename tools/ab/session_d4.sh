cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
for round in 1 2; do
  for v in d4 orig; do
    if [ $v = orig ]; then cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so; else cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so; fi
    echo "== $v round $round"; python tools/sponge_rate.py 2>/dev/null; python tools/sponge_rate.py --mixed 2>/dev/null
  done
done 2>&1 | tee gpurun_out/r03d/ab_d4.txt
cp tools/ab/libposeidon_d4.so sponge_amd/libposeidon_mi355x.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sponge or mixed or trace" ) 2>&1 | tail -2
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
