# round 5, session m: rows that only feed matrix-core inputs stay in operand form between the layers (t >= 4).  old = history rows only.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sponge_passes.py -x -q -m gpu -k "golden or widths or matrix_cores or default_table or c3 or mixed or odd_full or random_configs" ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
WORKLOADS="c3 w8 w7 w6 w5 w4 h9" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee $O/ab_operand_rows.txt
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
