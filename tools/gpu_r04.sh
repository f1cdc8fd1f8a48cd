#!/bin/bash
# Round-4 GPU-box session.  Usage (repo root on the GPU box): bash tools/gpu_r04.sh <tag> [stages]
# stages: any of  test smoke bench wide widepmc levels cov widths slots pmc prof   (default: "test smoke bench")
TAG=${1:-r04a}
STAGES=${2:-"test smoke bench"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
has() { [[ " $STAGES " == *" $1 "* ]]; }
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/device.txt
WIDE="--field bn254_fr --rate 8 --rounds 8 57 --log2 18"
if has test; then
  ( time timeout 3000 python -m pytest tests -x -q -m gpu --durations=15 ) > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit: $?" >> $OUT/pytest_gpu.log
fi
if has smoke; then timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; fi
if has cov; then PMX_COV_NO_BUILD=1 bash tools/mgpu_coverage.sh $OUT/mgpu_cov > $OUT/mgpu_cov.log 2>&1; fi
if has bench; then
  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
  timeout 600 python bench.py --workload c3 --steps 5 --warmup 1 --cpu-seconds 6 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
  timeout 600 python bench.py --workload c5 --total-log2 21 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c5_2e21.json 2> $OUT/bench_c5_2e21.err
  timeout 600 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
  timeout 600 python bench.py --workload c2 --total-log2 21 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_2e21.json 2> $OUT/bench_c2_2e21.err
  timeout 600 python bench.py --workload h3 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_h3.json 2> $OUT/bench_h3.err
  timeout 600 python bench.py --workload h9 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_h9.json 2> $OUT/bench_h9.err
  timeout 600 python bench.py --workload d3 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_d3.json 2> $OUT/bench_d3.err
  timeout 600 python bench.py --workload d9 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_d9.json 2> $OUT/bench_d9.err
fi
if has widths; then
  for w in w4 w5 w6 w7 w8; do timeout 600 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
fi
if has wide; then
  # the absorb / squeeze batch driver on wide states (BN254 Fr t = 9, and t = 7, 8 over BLS12-381 Fr), uniform and mixed modes
  ( timeout 300 python tools/sponge_rate.py $WIDE --absorb 8 --squeeze 8
    timeout 300 python tools/sponge_rate.py $WIDE --absorb 11 --squeeze 9
    timeout 300 python tools/sponge_rate.py $WIDE --absorb 11 --squeeze 9 --mixed
    timeout 300 python tools/sponge_rate.py --rate 7 --log2 18 --absorb 10 --squeeze 8
    timeout 300 python tools/sponge_rate.py --rate 7 --log2 18 --absorb 10 --squeeze 8 --mixed
    timeout 300 python tools/sponge_rate.py --rate 6 --log2 18 --absorb 9 --squeeze 7
    timeout 300 python tools/sponge_rate.py --rate 6 --log2 18 --absorb 9 --squeeze 7 --mixed
    timeout 300 python tools/sponge_rate.py --rate 5 --log2 18 --absorb 8 --squeeze 6 --mixed
    timeout 300 python tools/sponge_rate.py --rate 4 --log2 19 --absorb 7 --squeeze 5 --mixed
    timeout 300 python tools/sponge_rate.py --rate 3 --log2 19 --absorb 6 --squeeze 4 --mixed
    timeout 300 python tools/sponge_rate.py
    timeout 300 python tools/sponge_rate.py --mixed ) > $OUT/sponge_rate.txt 2>&1
fi
if has levels; then timeout 600 python tools/merkle_levels.py 21 > $OUT/merkle_levels.txt 2>&1; fi
cd /tmp && export TMPDIR=/tmp
if has widepmc; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_wide -- python3 $R/tools/sponge_rate.py $WIDE --absorb 11 --squeeze 9 --mixed > $OUT/prof_wide.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_wide -- python3 $R/tools/sponge_rate.py $WIDE --absorb 11 --squeeze 9 --mixed --reps 3 > $OUT/pmc_wide.log 2>&1
  f=$(find $OUT/prof_wide -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/wide_kernel_stats.csv
  python3 $R/tools/pmc_kernel_summary.py $OUT/pmc_wide > $OUT/wide_pmc_summary.txt 2>&1
fi
if has slots; then
  # SQ_INSTS_VALU of the dominant kernel of each bench workload -> profiles/valu_instructions.json entries
  for w in c2 c3 w4 w5 w6 w7 w8 h3 h9; do
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/slots_$w -- python3 $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/slots_$w.log 2>&1
  done
fi
if has pmc; then
  for w in c2 c3 h3 h9; do
    timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$w -- python3 $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_fetch_$w.log 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$w -- python3 $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_write_$w.log 2>&1
  done
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c2_2e21 -- python3 $R/bench.py --workload c2 --total-log2 21 --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_fetch_c2_2e21.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c2_2e21 -- python3 $R/bench.py --workload c2 --total-log2 21 --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_write_c2_2e21.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c5 -- python3 $R/bench.py --workload c5 --total-log2 21 --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_fetch_c5.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c5 -- python3 $R/bench.py --workload c5 --total-log2 21 --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_write_c5.log 2>&1
fi
if has prof; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_c2.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python3 $R/bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_c3.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 $R/bench.py --workload c5 --total-log2 21 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_c5.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_d9 -- python3 $R/bench.py --workload d9 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_d9.log 2>&1
  for w in c2 c3 c5 d9; do f=$(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${w}_kernel_stats.csv; done
fi
cd $R
if has slots; then
  J=$OUT/valu_instructions.json; rm -f $J
  python tools/valu_count.py $OUT/slots_c2 permute_kernel c2 1048576 "HybridEngine<3,5,mfma,windows of 3>" 3 $J "profiles/r04" > $OUT/valu_count.log 2>&1
  python tools/valu_count.py $OUT/slots_c3 permute_kernel c3 262144 "HybridEngine<9,5,mfma,windows of 6>" 2 $J "profiles/r04" >> $OUT/valu_count.log 2>&1
  python tools/valu_count.py $OUT/slots_w4 permute_kernel w4 524288 "HybridEngine<4,5,mfma,windows of 4>" 3 $J "profiles/r04" >> $OUT/valu_count.log 2>&1
  python tools/valu_count.py $OUT/slots_w5 permute_kernel w5 524288 "HybridEngine<5,5,mfma,windows of 5>" 3 $J "profiles/r04" >> $OUT/valu_count.log 2>&1
  python tools/valu_count.py $OUT/slots_w6 permute_kernel w6 262144 "HybridEngine<6,5,mfma,windows of 6>" 2 $J "profiles/r04" >> $OUT/valu_count.log 2>&1
  python tools/valu_count.py $OUT/slots_w7 permute_kernel w7 262144 "HybridEngine<7,5,mfma,windows of 6>" 2 $J "profiles/r04" >> $OUT/valu_count.log 2>&1
  python tools/valu_count.py $OUT/slots_w8 permute_kernel w8 262144 "HybridEngine<8,5,mfma,windows of 6>" 2 $J "profiles/r04" >> $OUT/valu_count.log 2>&1
  python tools/valu_count.py $OUT/slots_h3 hash_kernel h3 2097152 "HybridEngine<3,5,mfma,windows of 3>" 3 $J "profiles/r04" 2 >> $OUT/valu_count.log 2>&1
  python tools/valu_count.py $OUT/slots_h9 hash_kernel h9 262144 "HybridEngine<9,5,mfma,windows of 6>" 2 $J "profiles/r04" >> $OUT/valu_count.log 2>&1
  cat $OUT/valu_count.log
fi
if has pmc; then
  T=$OUT/hbm_traffic.json; rm -f $T
  python tools/extract_traffic.py $OUT/pmc_fetch_c2 $OUT/pmc_write_c2 permute_kernel c2 $T 1 1048576 > $OUT/traffic.log 2>&1
  python tools/extract_traffic.py $OUT/pmc_fetch_c3 $OUT/pmc_write_c3 permute_kernel c3 $T 1 262144 >> $OUT/traffic.log 2>&1
  python tools/extract_traffic.py $OUT/pmc_fetch_h3 $OUT/pmc_write_h3 hash_kernel h3 $T 1 2097152 >> $OUT/traffic.log 2>&1
  python tools/extract_traffic.py $OUT/pmc_fetch_h9 $OUT/pmc_write_h9 hash_kernel h9 $T 1 262144 >> $OUT/traffic.log 2>&1
  python tools/extract_traffic.py $OUT/pmc_fetch_c2_2e21 $OUT/pmc_write_c2_2e21 permute_kernel c2_2e21 $T 1 2097152 >> $OUT/traffic.log 2>&1
  python tools/extract_traffic.py $OUT/pmc_fetch_c5 $OUT/pmc_write_c5 compress c5 $T 21 2097151 >> $OUT/traffic.log 2>&1
  cat $OUT/traffic.log
fi
for f in sponge_rate merkle_levels wide_pmc_summary; do [ -f $OUT/$f.txt ] && cat $OUT/$f.txt; done
[ -f $OUT/wide_kernel_stats.csv ] && head -8 $OUT/wide_kernel_stats.csv | cut -c1-220
[ -f $OUT/mgpu_cov/coverage_summary.txt ] && head -40 $OUT/mgpu_cov/coverage_summary.txt
[ -f $OUT/pytest_gpu.log ] && tail -25 $OUT/pytest_gpu.log
[ -f $OUT/smoke.log ] && tail -1 $OUT/smoke.log
for f in $OUT/bench_*.json; do [ -f $f ] && python - <<PY
import json
try:
    d=json.load(open("$f"))
    vi = d.get("valu_issue") or {}
    print("$f".split("/")[-1], "%.4g perm/s"%d["value"], "ms/step %.3f"%d["ms_per_step"], "mad frac %.3f (peak %.3g, clk %.3g)"%(d["int_valu"]["frac"], d["int_valu"]["peak"], d["int_valu"]["shader_clock_hz"]), "issue frac", vi.get("frac"), "engine", (d.get("engine") or {}).get("name"), "verified", d["verified"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$f", "unreadable:", e); print(open("$f".replace(".json",".err")).read()[-1500:])
PY
done
