#!/bin/bash
# A/B of two builds of libposeidon_mi355x.so in one session (same device): interleaved rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp sponge_amd/libposeidon_mi355x.so /tmp/new.so
for round in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then cp tools/ab/libposeidon_old.so sponge_amd/libposeidon_mi355x.so; else cp /tmp/new.so sponge_amd/libposeidon_mi355x.so; fi
    python bench.py --workload ${W:-c2} --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v round $round %.4g perm/s  kernel_ms %.4f'%(d['value'],d['roofline']['kernel_ms']))"
  done
done
cp /tmp/new.so sponge_amd/libposeidon_mi355x.so
