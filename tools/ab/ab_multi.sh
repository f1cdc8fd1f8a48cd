#!/bin/bash
# A/B/C... of several builds in one session: VARIANTS="a b c" (tools/ab/libposeidon_<v>.so), WORKLOADS, STEPS, ROUNDS
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
for W in ${WORKLOADS:-c3}; do
for round in $(seq 1 ${ROUNDS:-3}); do
  for v in $VARIANTS; do
    cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so
    python bench.py --workload $W --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$W $v round $round %.4g perm/s  kernel_ms %.4f verified %s clk %.3g'%(d['value'],d['roofline']['kernel_ms'],d['verified'],d['int_valu']['shader_clock_hz']))"
  done
done
done
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
