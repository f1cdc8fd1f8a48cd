#!/bin/bash
# A/B of two builds of libposeidon_mi355x.so in one session (same device), interleaved rounds:
# tools/ab/libposeidon_old.so vs tools/ab/libposeidon_new.so (both git-ignored).  Runs on the GPU box's snapshot;
# the tree's own library is put back at the end.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
for W in ${WORKLOADS:-c2}; do
for round in 1 2 3; do
  for v in new old; do
    cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so
    python bench.py --workload $W --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$W $v round $round %.4g perm/s  kernel_ms %.4f'%(d['value'],d['roofline']['kernel_ms']))"
  done
done
done
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
