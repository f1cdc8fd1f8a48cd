#!/bin/bash
# A/B of window sizes of the partial rounds in one session on one device: tools/ab/libposeidon_win<K>.so, three rounds, interleaved.
#   VARIANTS="win0 win3 win4 win6" WORKLOADS="c3 w7 w8" bash tools/ab/session_windows.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
for W in ${WORKLOADS:-c3 w7 w8}; do
for round in 1 2 3; do
  for v in ${VARIANTS:-win3 win4 win6}; do
    [ -f tools/ab/libposeidon_$v.so ] || continue
    cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so
    python bench.py --workload $W --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$W $v round $round %.4g perm/s  kernel_ms %.4f  %s verified %s'%(d['value'],d['roofline']['kernel_ms'],d['engine']['name'],d['verified']))"
  done
done
done
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
