# round 5, session e: row finish without the v_mov of the correction words (weights in vector registers), funnel-shift re-cut; LDS read-ahead
# per width.  Full GPU suite on the new build, A/B against the round's earlier build over every width, t = 3 at four waves per SIMD as a variant.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
( time timeout 3000 python -m pytest tests -x -q -m gpu --durations=10 ) > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
WORKLOADS="c3 w8 w7 w6 w5 w4 c2 h3 h9 k3" STEPS=10 bash tools/ab/ab.sh 2>&1 | tee $O/ab_row_finish.txt
cp tools/ab/libposeidon_new.so tools/ab/libposeidon_old.so; cp tools/ab/libposeidon_t3w4.so tools/ab/libposeidon_new.so
WORKLOADS="c2 h3 k3" STEPS=20 bash tools/ab/ab.sh 2>&1 | sed 's/ new / t3w4 /; s/ old / t3w3 /' | tee $O/ab_t3_four_waves.txt
