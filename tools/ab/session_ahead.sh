# round 5, session d: the A operand of a stage read out of the LDS tile ahead of the products (pmx_mfma.hpp: PMX_MFMA_LDS_AHEAD = 8).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or widths or matrix_cores or default_table or ragged or odd_full" ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
WORKLOADS="c3 w8 w7 w6 w5 w4 c2 h9" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee $O/ab_lds_ahead.txt
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
bash tools/pmc_c3_stalls.sh c3 $GRAFT_REPO_ROOT/$O/stalls_c3 > $O/stalls_c3.txt 2>&1; cat $O/stalls_c3.txt
