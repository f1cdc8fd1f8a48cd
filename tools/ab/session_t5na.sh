# round 5, session p: t = 5 without fetching the history rows' tables ahead (no spill) against fetched ahead (32 bytes of scratch)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p; mkdir -p $O
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "default_table_widths or golden or random_configs" ) 2>&1 | tail -1
WORKLOADS="w5" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee $O/ab_t5_no_fetch_ahead.txt
