# round 5, session i: 32 bytes (one k-step) per element in the matrix-core layers, balanced stages, fetch buffers per stage.  old = session e's build.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
WORKLOADS="c3 w8 w7 w6 w5 w4 c2 h3 h9 k3" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee $O/ab_32_byte_elements.txt
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
