# round 5, session b: the row finish of layer row i - 1 issued beside the matrix-core products of row i (pmx_mfma.hpp: PMX_MFMA_PIPELINE).
# parity first, then A/B old (same source, PMX_MFMA_PIPELINE as before) vs new over every width; then the launcher-free bench tests.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or widths or matrix_cores or default_table or ragged or odd_full" ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
( timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sponge_passes.py -x -q -m gpu -k "c3 or c2 or t3 or mixed" ) > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log
WORKLOADS="c3 w8 w7 w6 w5 w4 c2 h9" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee $O/ab_pipeline.txt
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
( timeout 2400 python -m pytest tests/test_gpu_mgpu_standin.py tests/test_gpu_mgpu.py -x -q -m gpu ) > $O/pytest_mgpu.log 2>&1; tail -15 $O/pytest_mgpu.log
