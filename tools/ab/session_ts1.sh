cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03p
cp sponge_amd/libposeidon_mi355x.so /tmp/tree.so
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
( timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or ragged or merkle" ) 2>&1 | tail -2
cp /tmp/tree.so sponge_amd/libposeidon_mi355x.so
WORKLOADS="c2 h3" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee gpurun_out/r03p/ab_ts1.txt
WORKLOADS="c5" STEPS=10 BENCH_ARGS="--total-log2 21" bash tools/ab/ab.sh 2>&1 | tee -a gpurun_out/r03p/ab_ts1.txt
