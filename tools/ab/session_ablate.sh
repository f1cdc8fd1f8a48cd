# round 5, session h: where does a t = 9 wave wait?  Timing-only ablations of the matrix-core layer (results are garbage, --no-verify):
# abl1 = no workgroup barriers, abl2 = the A operand not read from LDS, abl3 = no matrix-core products (a VALU add in their place).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
for round in 1 2; do for v in new abl1 abl2 abl3; do
  cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so
  for l in 16 18; do
    python bench.py --workload c3 --total-log2 $l --steps 20 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v c3 2^$l states: %.4g perm/s  kernel_ms %.4f'%(d['value'],d['roofline']['kernel_ms']))"
  done
done; done 2>&1 | tee $O/c3_ablations.txt
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
