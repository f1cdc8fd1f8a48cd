cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03o
cp sponge_amd/libposeidon_mi355x.so /tmp/tree.so
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
( timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "widths or golden or ragged or run_time or odd or random or sponge or mixed" ) 2>&1 | tail -2
cp /tmp/tree.so sponge_amd/libposeidon_mi355x.so
WORKLOADS="c3 w6 w7 w8" STEPS=8 bash tools/ab/ab.sh 2>&1 | tee gpurun_out/r03o/ab_rio.txt
