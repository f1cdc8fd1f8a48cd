cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03v
cp sponge_amd/libposeidon_mi355x.so /tmp/tree.so
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
( timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "widths or golden or ragged or random" ) 2>&1 | tail -2
cp /tmp/tree.so sponge_amd/libposeidon_mi355x.so
WORKLOADS="c3 w6 w7 w8 h9" STEPS=8 bash tools/ab/ab.sh 2>&1 | tee gpurun_out/r03v/ab_r0t.txt
