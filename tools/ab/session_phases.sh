# round 6, session b: per-phase time of the t = 9 window kernel (diagnostic build, tools/diag/phase_ticks.py), full launch and a lone-wave launch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
cp tools/ab/libposeidon_phases.so sponge_amd/libposeidon_mi355x.so
( python tools/diag/phase_ticks.py 18 9; python tools/diag/phase_ticks.py 16 9; python tools/diag/phase_ticks.py 18 8 ) 2>&1 | tee $O/phase_ticks.txt
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
