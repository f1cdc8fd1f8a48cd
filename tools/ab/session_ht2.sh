# round 5, session n: t = 3 with its one history term per window as a matrix-core row as well (and the second carried lane in operand form):
# base = shifted table (shipped), ht2 = rows at four waves per SIMD (28 bytes of scratch), ht2w3 = rows at three waves
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
cp sponge_amd/libposeidon_mi355x.so tools/ab/libposeidon_base.so
for v in ht2 ht2w3; do
  cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so
  ( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sponge_passes.py -x -q -m gpu -k "golden or t3 or merkle_trees" ) 2>&1 | tail -1
done | tee $O/pytest.txt
for W in c2 h3 c5s; do for round in 1 2 3; do for v in ht2 ht2w3 base; do
  cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so
  case $W in c5s) args="--workload c5 --total-log2 21 --steps 10";; *) args="--workload $W --steps 20";; esac
  python bench.py $args --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$W $v round $round %.4g perm/s  kernel_ms %.4f'%(d['value'],d['roofline']['kernel_ms']))"
done; done; done 2>&1 | tee $O/ab_t3_history_row.txt
cp tools/ab/libposeidon_base.so sponge_amd/libposeidon_mi355x.so
