cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mgpu.py -x -q -m gpu -k "sponge or mixed or trace or absorb or hash or widths" ) > gpurun_out/r03c/pytest_sponge.log 2>&1; tail -3 gpurun_out/r03c/pytest_sponge.log
python tools/sponge_rate.py > gpurun_out/r03c/sponge_rate.txt 2>&1; python tools/sponge_rate.py --mixed >> gpurun_out/r03c/sponge_rate.txt 2>&1; cat gpurun_out/r03c/sponge_rate.txt
WORKLOADS="w4 w5 w6" STEPS=8 bash tools/ab/ab.sh 2>&1 | tee gpurun_out/r03c/ab_w4.txt
