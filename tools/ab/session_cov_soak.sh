# round 5, session o: gcov line coverage of pmx_mgpu.cpp under the stand-in tests with the test-hook build, and a soak of the final build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
PMX_COV_NO_BUILD=1 bash tools/mgpu_coverage.sh $O/mgpu_cov > $O/mgpu_cov.log 2>&1; head -12 $O/mgpu_cov/coverage_summary.txt
timeout 900 python tools/soak.py 420 > $O/soak.txt 2>&1; tail -3 $O/soak.txt
