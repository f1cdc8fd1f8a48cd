# round 5, session k: the history terms of the window S-box inputs as rows on the matrix cores (t >= 6).  old = the y build (hist on the VALU),
# new = rows issued behind their S-box, hr1 = rows issued in front of it (spills at t = 8, 9).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sponge_passes.py -x -q -m gpu -k "golden or widths or matrix_cores or default_table or c3 or mixed or odd_full or random_configs" ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
WORKLOADS="c3 w8 w7 w6 h9 d9" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee $O/ab_history_rows.txt
cp tools/ab/libposeidon_new.so /tmp/keep_new.so; cp tools/ab/libposeidon_new.so tools/ab/libposeidon_old.so; cp tools/ab/libposeidon_hr1.so tools/ab/libposeidon_new.so
WORKLOADS="c3 w8 w7 w6" STEPS=10 bash tools/ab/ab.sh 2>&1 | sed 's/ new / hr1(front) /; s/ old / hr3(behind) /' | tee $O/ab_history_rows_order.txt
cp /tmp/keep_new.so sponge_amd/libposeidon_mi355x.so
