#!/bin/bash
# Round-3 GPU-box session.  Usage (repo root on the GPU box): bash tools/gpu_r03.sh <tag> [stages]
# stages: any of  test smoke bench prof pmc slots sponge levels lone paths   (default: "test smoke bench")
TAG=${1:-r03a}
STAGES=${2:-"test smoke bench"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
has() { [[ " $STAGES " == *" $1 "* ]]; }
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/device.txt
if has test; then
  ( time timeout 3000 python -m pytest tests -x -q -m gpu --durations=15 ) > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit: $?" >> $OUT/pytest_gpu.log
fi
if has smoke; then timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; fi
if has bench; then
  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
  timeout 600 python bench.py --workload c3 --steps 5 --warmup 1 --cpu-seconds 6 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
  timeout 600 python bench.py --workload c5 --total-log2 21 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c5_2e21.json 2> $OUT/bench_c5_2e21.err
  timeout 600 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
  timeout 600 python bench.py --workload c2 --total-log2 21 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_2e21.json 2> $OUT/bench_c2_2e21.err
  timeout 600 python bench.py --workload h3 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_h3.json 2> $OUT/bench_h3.err
  timeout 600 python bench.py --workload h9 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_h9.json 2> $OUT/bench_h9.err
fi
if has sponge; then
  timeout 300 python tools/sponge_rate.py > $OUT/sponge_rate.txt 2>&1
  timeout 300 python tools/sponge_rate.py --mixed >> $OUT/sponge_rate.txt 2>&1
fi
if has levels; then timeout 600 python tools/merkle_levels.py 21 > $OUT/merkle_levels.txt 2>&1; fi
if has paths; then timeout 600 python tools/verify_paths_rate.py 15 24 > $OUT/verify_paths.txt 2>&1; timeout 600 python tools/verify_paths_rate.py 10 24 >> $OUT/verify_paths.txt 2>&1; fi
if has lone; then timeout 300 tools/lone_wave_microbench > $OUT/lone_wave_microbench.txt 2>&1; fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_c2.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python3 $R/bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_c3.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 $R/bench.py --workload c5 --total-log2 21 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_c5.log 2>&1
fi
if has sponge; then
  # the mid-stream driver kernels (absorb_kernel / squeeze_kernel): per-kernel time, then VALU instruction counts in their own pass
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sponge -- python3 $R/tools/sponge_rate.py > $OUT/prof_sponge.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sponge_mixed -- python3 $R/tools/sponge_rate.py --mixed > $OUT/prof_sponge_mixed.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sponge -- python3 $R/tools/sponge_rate.py > $OUT/pmc_sponge.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sponge_mixed -- python3 $R/tools/sponge_rate.py --mixed > $OUT/pmc_sponge_mixed.log 2>&1
fi
if has slots; then
  # VALU issue-slot accounting of the dominant kernels: instruction counts and SQ cycle shares, in two counter passes
  for w in c2 c3; do
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/slots_a_$w -- python3 $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/slots_a_$w.log 2>&1
    timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/slots_b_$w -- python3 $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/slots_b_$w.log 2>&1
  done
fi
if has pmc; then
  for w in c2 c3; do
    timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$w -- python3 $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_fetch_$w.log 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$w -- python3 $R/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_write_$w.log 2>&1
  done
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c5 -- python3 $R/bench.py --workload c5 --total-log2 21 --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_fetch_c5.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c5 -- python3 $R/bench.py --workload c5 --total-log2 21 --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_write_c5.log 2>&1
fi
cd $R
if has pmc; then
  python tools/extract_traffic.py $OUT/pmc_fetch_c2 $OUT/pmc_write_c2 permute_kernel c2 $OUT/hbm_traffic.json 1 1048576 > $OUT/traffic.log 2>&1
  python tools/extract_traffic.py $OUT/pmc_fetch_c3 $OUT/pmc_write_c3 permute_kernel c3 $OUT/hbm_traffic.json 1 262144 >> $OUT/traffic.log 2>&1
  python tools/extract_traffic.py $OUT/pmc_fetch_c5 $OUT/pmc_write_c5 compress c5 $OUT/hbm_traffic.json 21 2097151 >> $OUT/traffic.log 2>&1
  cat $OUT/traffic.log
fi
if has slots; then
  ( python tools/valu_slots.py $OUT/slots_a_c2 permute_kernel 1048576 1 40077; python tools/valu_slots.py $OUT/slots_b_c2 permute_kernel 1048576 1 40077
    python tools/valu_slots.py $OUT/slots_a_c3 permute_kernel 262144 1 146160; python tools/valu_slots.py $OUT/slots_b_c3 permute_kernel 262144 1 146160 ) > $OUT/valu_slots.txt 2>&1
  cat $OUT/valu_slots.txt
fi
if has sponge; then
  python tools/sponge_pmc_summary.py $OUT > $OUT/sponge_summary.txt 2>&1; cat $OUT/sponge_summary.txt
fi
if has prof; then
  for w in c2 c3 c5; do f=$(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${w}_kernel_stats.csv && head -8 $f | cut -c1-200; done
fi
for f in sponge_rate merkle_levels verify_paths lone_wave_microbench; do [ -f $OUT/$f.txt ] && cat $OUT/$f.txt; done
[ -f $OUT/pytest_gpu.log ] && tail -25 $OUT/pytest_gpu.log
[ -f $OUT/smoke.log ] && tail -1 $OUT/smoke.log
for f in $OUT/bench_*.json; do [ -f $f ] && python - <<PY
import json
try:
    d=json.load(open("$f"))
    print("$f".split("/")[-1], "%.4g perm/s"%d["value"], "ms/step %.3f"%d["ms_per_step"], "mad frac %.3f (peak %.3g, clk %.3g)"%(d["int_valu"]["frac"], d["int_valu"]["peak"], d["int_valu"]["shader_clock_hz"]), "verified", d["verified"], "cpu", (d.get("cpu_baseline") or {}).get("value"), ((d.get("cpu_baseline") or {}).get("single_thread") or {}).get("value"))
except Exception as e:
    print("$f", "unreadable:", e); print(open("$f".replace(".json",".err")).read()[-1500:])
PY
done
