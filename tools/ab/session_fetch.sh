# round 5, session g: per-stage fetch buffers refilled a row ahead (t = 9: two stages), LDS read-ahead 4 at t = 9.  old = session e's build.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or widths or matrix_cores or default_table" ) > $O/pytest_parity.log 2>&1; tail -2 $O/pytest_parity.log
WORKLOADS="c3 h9 w8 c2" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee $O/ab_fetch_a_row_ahead.txt
for v in old new; do cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so
for l in 16 17; do
  python bench.py --workload c3 --total-log2 $l --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v c3 2^$l states: %.4g perm/s  kernel_ms %.4f'%(d['value'],d['roofline']['kernel_ms']))"
done; done 2>&1 | tee $O/c3_lone_workgroup.txt
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
