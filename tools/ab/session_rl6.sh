cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03w
cp sponge_amd/libposeidon_mi355x.so /tmp/tree.so
( timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "widths or golden or ragged or random or odd or sponge" ) 2>&1 | tail -2
WORKLOADS="w6 c3" STEPS=8 bash tools/ab/ab.sh 2>&1 | tee gpurun_out/r03w/ab_rl6.txt
