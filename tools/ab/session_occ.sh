# round 5, session f: does the second wave on a SIMD buy anything at t = 9?  One workgroup per CU (2^16 states = 256 workgroups) against two
# (2^17) and two rounds of two (2^18); and the VALU instruction counts of the two builds of session e.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so    # (the build with the slimmer row finish)
for l in 15 16 17 18 19 20; do
  python bench.py --workload c3 --total-log2 $l --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c3 2^$l states: %.4g perm/s  kernel_ms %.4f  clk %.3g'%(d['value'],d['roofline']['kernel_ms'],d['int_valu']['shader_clock_hz']))"
done 2>&1 | tee $O/c3_by_batch_size.txt
for l in 17 18 19 20 21; do
  python bench.py --workload c2 --total-log2 $l --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c2 2^$l states: %.4g perm/s  kernel_ms %.4f'%(d['value'],d['roofline']['kernel_ms']))"
done 2>&1 | tee $O/c2_by_batch_size.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in new r04head; do
  cp $R/tools/ab/libposeidon_$v.so $R/sponge_amd/libposeidon_mi355x.so
  for w in c3 c2; do
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $R/$O/slots_${v}_$w -- python3 $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $R/$O/slots_${v}_$w.log 2>&1
    echo "== $v $w"; python3 $R/tools/pmc_kernel_summary.py $R/$O/slots_${v}_$w 2>&1 | grep -A14 "permute_kernel" | head -16
  done
done 2>&1 | tee $R/$O/valu_counts.txt
cp $R/tools/ab/libposeidon_new.so $R/sponge_amd/libposeidon_mi355x.so
