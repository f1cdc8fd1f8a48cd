cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03m
cp sponge_amd/libposeidon_mi355x.so /tmp/tree.so
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
( timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > gpurun_out/r03m/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03m/pytest_gpu.log | tail -3
cp /tmp/tree.so sponge_amd/libposeidon_mi355x.so
WORKLOADS="c3 w4 w5 w6 w7 w8" STEPS=8 bash tools/ab/ab.sh 2>&1 | tee gpurun_out/r03m/ab_dio.txt
