cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "widths or golden or ragged or run_time or odd" ) 2>&1 | tail -2
WORKLOADS="c3 h9 w7 w8" STEPS=8 bash tools/ab/ab.sh 2>&1 | tee gpurun_out/r03f/ab_mu2.txt
