cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03i
cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
for W in c3 w6 w7 w8 h9; do
for round in 1 2; do
  for v in wn0 tree old; do
    if [ $v = tree ]; then cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so; else cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so; fi
    python bench.py --workload $W --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$W $v round $round %.4g perm/s  kernel_ms %.4f verified %s'%(d['value'],d['roofline']['kernel_ms'],d['verified']))"
  done
done
done 2>&1 | tee gpurun_out/r03i/ab_wn.txt
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
