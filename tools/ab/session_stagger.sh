# round 6, session a: do the two workgroups of a CU run in lockstep?  tools/hwid_probe (who shares a CU, what tells them apart), then the
# t = 6 ... 9 window kernels with one of the two held back at the start (PMX_STAGGER = "sleeps of 127 x 64 clocks, source of the bit":
# 0 wave slot parity, 1 workgroup slot parity, 2 second half of the first generation's blocks, 3 bit 3 of the block index).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
./tools/hwid_probe 2048 3000 > $O/hwid_probe.txt 2>&1; head -12 $O/hwid_probe.txt; tail -4 $O/hwid_probe.txt
run() {  # workload, stagger
  PMX_STAGGER=$2 python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 stagger=$2 %.4g perm/s  kernel_ms %.4f verified %s clk %.3g'%(d['value'],d['roofline']['kernel_ms'],d['verified'],d['int_valu']['shader_clock_hz']))"
}
for round in 1 2; do
  for st in 0 2,0 4,0 6,0 8,0 12,0 4,1 8,1 4,2 8,2 4,3 8,3; do run c3 $st; done
done 2>&1 | tee $O/stagger_c3.txt
for w in w6 w7 w8 h9; do for st in 0 4,1 8,1 0; do run $w $st; done; done 2>&1 | tee $O/stagger_widths.txt
