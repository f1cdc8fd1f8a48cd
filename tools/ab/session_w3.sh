cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mgpu.py -x -q -m gpu -k "sponge or mixed or trace or absorb or hash" ) > gpurun_out/r03b/pytest_sponge.log 2>&1; tail -3 gpurun_out/r03b/pytest_sponge.log
python tools/sponge_rate.py > gpurun_out/r03b/sponge_rate.txt 2>&1; python tools/sponge_rate.py --mixed >> gpurun_out/r03b/sponge_rate.txt 2>&1; cat gpurun_out/r03b/sponge_rate.txt
WORKLOADS="c3 w6 w7 w8 h9" STEPS=8 bash tools/ab/ab.sh 2>&1 | tee gpurun_out/r03b/ab_w3.txt
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "widths or golden or ragged" ) 2>&1 | tail -2
