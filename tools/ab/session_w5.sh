# round 5, session q: the w5 line, its counters, and the full GPU suite once more on the final build (t = 5 without fetch-ahead)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --workload w5 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_w5.json 2> $O/bench_w5.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --workload c3 --steps 5 --warmup 1 --cpu-seconds 6 > $O/bench_c3.json 2> $O/bench_c3.err
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $R/$O/slots_w5 -- python3 $R/bench.py --workload w5 --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $R/$O/slots_w5.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch_w5 -- python3 $R/bench.py --workload w5 --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $R/$O/pmc_fetch_w5.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write_w5 -- python3 $R/bench.py --workload w5 --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $R/$O/pmc_write_w5.log 2>&1
cd $R
python tools/valu_count.py $O/slots_w5 permute_kernel w5 524288 "HybridEngine<5,5,mfma,windows of 5>" 3 $O/valu_w5.json "profiles/r05"
python tools/extract_traffic.py $O/pmc_fetch_w5 $O/pmc_write_w5 permute_kernel w5 $O/traffic_w5.json 1 524288
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], '%.4g'%d['value'], d['verified'], (d.get('valu_issue') or {}).get('frac'))"; done
