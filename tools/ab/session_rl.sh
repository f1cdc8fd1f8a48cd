cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s
mkdir -p $OUT
( timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "widths or golden or ragged or c3 or odd or random or sponge or mixed" ) 2>&1 | tail -2
WORKLOADS="c3" STEPS=8 bash tools/ab/ab.sh 2>&1 | tee $OUT/ab_rl.txt
cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
for round in 1 2; do
  for v in d4 orig; do
    if [ $v = orig ]; then cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so; else cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so; fi
    echo "== $v round $round"; python tools/sponge_rate.py 2>/dev/null; python tools/sponge_rate.py --mixed 2>/dev/null
  done
done 2>&1 | tee $OUT/ab_d4.txt
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c3 -- python3 $R/bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_fetch_c3.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c3 -- python3 $R/bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_write_c3.log 2>&1
cd $R
python tools/extract_traffic.py $OUT/pmc_fetch_c3 $OUT/pmc_write_c3 permute_kernel c3 $OUT/hbm_traffic.json 1 262144
