# round 5, session l: history rows on the matrix cores at t = 4, 5 as well (PMX_MFMA_HIST_TAB_MAX_T = 3) against shifted tables there (5)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "default_table_widths or golden" ) > $O/pytest_parity.log 2>&1; tail -2 $O/pytest_parity.log
WORKLOADS="w4 w5" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee $O/ab_history_rows_t4_t5.txt
