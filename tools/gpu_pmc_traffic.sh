#!/bin/bash
# only the HBM-traffic PMC passes of gpu_round.sh (c3, c5), merged into an existing hbm_traffic.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-traffic}
mkdir -p $OUT
cp $R/profiles/hbm_traffic.json $OUT/hbm_traffic.json
cd /tmp && export TMPDIR=/tmp
for w in c3 c5; do
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_$w.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_$w.log 2>&1
done
cd $R
python tools/extract_traffic.py $OUT/pmc_fetch_c3 $OUT/pmc_write_c3 permute_kernel c3 $OUT/hbm_traffic.json
python tools/extract_traffic.py $OUT/pmc_fetch_c5 $OUT/pmc_write_c5 compress c5 $OUT/hbm_traffic.json 21
