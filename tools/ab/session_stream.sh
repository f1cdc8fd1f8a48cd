# round 6, session c: the matrix-core rows WITHOUT the LDS tile and its workgroup barriers - every wave streams the A operand of its rows from
# global memory into a ring of registers (pmx_mfma.hpp: PMX_MFMA_STREAM).  Parity subset on the new build first, then A/B over every width.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
cp sponge_amd/libposeidon_mi355x.so /tmp/keep.so
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or widths or matrix_cores or default_table or ragged or odd_full" ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
cp /tmp/keep.so sponge_amd/libposeidon_mi355x.so
WORKLOADS="${WL:-c3 w8 w7 w6 w5 w4 c2 h9 d9}" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee $O/ab_stream.txt
