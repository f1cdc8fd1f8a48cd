#!/bin/bash
# One GPU-box session: parity tests, smoke, bench lines, rocprofv3 kernel stats and PMC traffic passes.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag] [quick]
TAG=${1:-r01}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/device.txt
if [ -z "$QUICK" ]; then
  ( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit: $?" >> $OUT/pytest_gpu.log
else
  ( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit: $?" >> $OUT/pytest_gpu.log
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
timeout 600 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 600 python bench.py --workload c3 --steps 5 --warmup 1 --cpu-seconds 6 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
timeout 600 python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
timeout 600 python bench.py --workload h3 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_h3.json 2> $OUT/bench_h3.err
timeout 600 python bench.py --workload h9 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_h9.json 2> $OUT/bench_h9.err
timeout 300 python tools/host_path_rate.py > $OUT/host_path.json 2> $OUT/host_path.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python3 $R/bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_c3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 $R/bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_c5.log 2>&1
# HBM traffic: counters in their own passes, one counter block per pass (no trace/stats domains mixed in)
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c2 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_c2.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c2 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_c2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c3 -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_c3.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c3 -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_c3.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c5 -- python3 $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_c5.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c5 -- python3 $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_c5.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_c2 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq_c2.log 2>&1
cd $R
python tools/extract_traffic.py $OUT/pmc_fetch_c2 $OUT/pmc_write_c2 permute_kernel c2 $OUT/hbm_traffic.json > $OUT/traffic.log 2>&1
python tools/extract_traffic.py $OUT/pmc_fetch_c3 $OUT/pmc_write_c3 permute_kernel c3 $OUT/hbm_traffic.json >> $OUT/traffic.log 2>&1
python tools/extract_traffic.py $OUT/pmc_fetch_c5 $OUT/pmc_write_c5 compress c5 $OUT/hbm_traffic.json 21 >> $OUT/traffic.log 2>&1
tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; cat $OUT/bench_c2.json $OUT/bench_c3.json $OUT/bench_c5.json $OUT/bench_h3.json $OUT/bench_h9.json | cut -c1-420; cat $OUT/host_path.json; cat $OUT/traffic.log
