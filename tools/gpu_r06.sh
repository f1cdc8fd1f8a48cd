#!/bin/bash
# Round-6 GPU-box session.  Usage (repo root on the GPU box): bash tools/gpu_r06.sh <tag> [stages]
# stages: any of  test testlib fuzz smoke bench widths prof slots pmc hostpath rehearsal rates stalls   (default: "test smoke bench")
# Order inside a session: tests, the counter passes (prof slots pmc stalls) and their JSONs, THEN the bench lines - which therefore
# carry this lease's own valu_issue / traffic figures (evidence of one build from one lease at one clock).
TAG=${1:-r06z}
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
if has testlib; then   # the same suite once more with libposeidon_mi355x_test.so bound for the whole session (the shipped objects + the device-group hooks)
  ( time timeout 3000 python -m pytest tests -x -q -m gpu --pmx-test-library ) > $OUT/pytest_gpu_test_library.log 2>&1
  echo "pytest exit: $?" >> $OUT/pytest_gpu_test_library.log
fi
if has fuzz; then ( timeout 1500 python tools/diag/fuzz_configs.py 600 501 --matrix; timeout 900 python tools/diag/alpha1_widths.py | grep -c "pairs: 0 of" ) > $OUT/fuzz_configs.txt 2>&1; fi
if has smoke; then timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; fi
# ---- the counter passes come FIRST: the bench lines of this session then quote the instruction counts and traffic of this very lease
cd /tmp && export TMPDIR=/tmp
PMC_ARGS() {  # workload -> bench arguments of its counter passes
  case $1 in
    c2_2e21) echo "--workload c2 --total-log2 21";;
    c5_2e21) echo "--workload c5 --total-log2 21";;
    *) echo "--workload $1";;
  esac
}
if has prof; then
  for w in c2 c3 c5_2e21 d9; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -- python3 $R/bench.py $(PMC_ARGS $w) --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_$w.log 2>&1
    f=$(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${w}_kernel_stats.csv
  done
fi
if has slots; then
  for w in c2 c3 k3 w4 w5 w6 w7 w8 h3 h9 d3 d9 c5_2e21; do
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/slots_$w -- python3 $R/bench.py $(PMC_ARGS $w) --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/slots_$w.log 2>&1
  done
fi
if has pmc; then
  for w in c2 c3 k3 h3 h9 d3 d9 w4 w5 w6 w7 w8 c2_2e21 c5_2e21 c5; do
    steps=4; [ $w = c5 ] && steps=2
    timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$w -- python3 $R/bench.py $(PMC_ARGS $w) --steps $steps --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_fetch_$w.log 2>&1
    timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$w -- python3 $R/bench.py $(PMC_ARGS $w) --steps $steps --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_write_$w.log 2>&1
  done
fi
if has stalls; then bash $R/tools/pmc_c3_stalls.sh c3 $OUT/stalls_c3 > $OUT/pmc_c3_stalls.txt 2>&1; bash $R/tools/pmc_c3_stalls.sh c2 $OUT/stalls_c2 > $OUT/pmc_c2_stalls.txt 2>&1; fi
cd $R
if has slots; then
  J=$OUT/valu_instructions.json; rm -f $J
  V() { python tools/valu_count.py "$@" >> $OUT/valu_count.log 2>&1; }
  : > $OUT/valu_count.log
  V $OUT/slots_c2 permute_kernel c2 1048576 "HybridEngine<3,5,mfma,windows of 3>" 4 $J "profiles/r06"
  V $OUT/slots_k3 permute_kernel k3 1048576 "HybridEngine<3,0,mfma,windows of 3>" 4 $J "profiles/r06"
  V $OUT/slots_c3 permute_kernel c3 262144 "HybridEngine<9,5,mfma,windows of 9>" 2 $J "profiles/r06"
  V $OUT/slots_w4 permute_kernel w4 524288 "HybridEngine<4,5,mfma,windows of 4>" 3 $J "profiles/r06"
  V $OUT/slots_w5 permute_kernel w5 524288 "HybridEngine<5,5,mfma,windows of 5>" 3 $J "profiles/r06"
  V $OUT/slots_w6 permute_kernel w6 262144 "HybridEngine<6,5,mfma,windows of 6>" 2 $J "profiles/r06"
  V $OUT/slots_w7 permute_kernel w7 262144 "HybridEngine<7,5,mfma,windows of 7>" 2 $J "profiles/r06"
  V $OUT/slots_w8 permute_kernel w8 262144 "HybridEngine<8,5,mfma,windows of 8>" 2 $J "profiles/r06"
  V $OUT/slots_h3 hash_kernel h3 2097152 "HybridEngine<3,5,mfma,windows of 3>" 4 $J "profiles/r06" 2
  V $OUT/slots_h9 hash_kernel h9 262144 "HybridEngine<9,5,mfma,windows of 9>" 2 $J "profiles/r06"
  cat $OUT/valu_count.log
  cp $J $R/profiles/valu_instructions.json   # (on the box: the bench lines below quote the counts of THIS session, same lease, same clock)
  for w in d3 d9 c5_2e21; do echo "== $w (every kernel of the step)"; python3 tools/pmc_kernel_summary.py $OUT/slots_$w 2>&1 | head -60; done > $OUT/valu_driver_and_tree_kernels.txt
fi
if has pmc; then
  T=$OUT/hbm_traffic.json; rm -f $T
  X() { python tools/extract_traffic.py "$@" >> $OUT/traffic.log 2>&1; }
  : > $OUT/traffic.log
  X $OUT/pmc_fetch_c2 $OUT/pmc_write_c2 permute_kernel c2 $T 1 1048576
  X $OUT/pmc_fetch_k3 $OUT/pmc_write_k3 permute_kernel k3 $T 1 1048576
  X $OUT/pmc_fetch_c3 $OUT/pmc_write_c3 permute_kernel c3 $T 1 262144
  X $OUT/pmc_fetch_h3 $OUT/pmc_write_h3 hash_kernel h3 $T 1 2097152
  X $OUT/pmc_fetch_h9 $OUT/pmc_write_h9 hash_kernel h9 $T 1 262144
  for w in w4 w5; do X $OUT/pmc_fetch_$w $OUT/pmc_write_$w permute_kernel $w $T 1 524288; done
  for w in w6 w7 w8; do X $OUT/pmc_fetch_$w $OUT/pmc_write_$w permute_kernel $w $T 1 262144; done
  X $OUT/pmc_fetch_c2_2e21 $OUT/pmc_write_c2_2e21 permute_kernel c2_2e21 $T 1 2097152
  X $OUT/pmc_fetch_c5_2e21 $OUT/pmc_write_c5_2e21 compress c5_2e21 $T 21 2097151
  X $OUT/pmc_fetch_c5 $OUT/pmc_write_c5 compress c5 $T 24 16777215
  # the duplex driver: a step = one absorb call + one squeeze call = 4 kernels (sponge_first x 2, permute_listed x 2) at d9, the same at d3
  X $OUT/pmc_fetch_d3 $OUT/pmc_write_d3 "<pmx::HybridEngine" d3 $T 4 4194304
  X $OUT/pmc_fetch_d9 $OUT/pmc_write_d9 "<pmx::HybridEngine" d9 $T 4 1048576
  cat $OUT/traffic.log
  cp $T $R/profiles/hbm_traffic.json
fi
cd $R
B() { local name=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
if has bench; then
  B c2 --steps 20 --warmup 5
  B c3 --workload c3 --steps 5 --warmup 1 --cpu-seconds 6
  B c5_2e21 --workload c5 --total-log2 21 --steps 10 --warmup 2 --no-cpu-baseline
  B c5 --workload c5 --steps 3 --warmup 1 --no-cpu-baseline
  B c2_2e21 --workload c2 --total-log2 21 --steps 10 --warmup 2 --no-cpu-baseline
  B k3 --workload k3 --steps 20 --warmup 5 --no-cpu-baseline
  B h3 --workload h3 --steps 10 --warmup 2 --no-cpu-baseline
  B h9 --workload h9 --steps 5 --warmup 1 --no-cpu-baseline
  B d3 --workload d3 --steps 10 --warmup 2 --no-cpu-baseline
  B d9 --workload d9 --steps 5 --warmup 2 --no-cpu-baseline
fi
if has widths; then for w in w4 w5 w6 w7 w8; do B $w --workload $w --steps 5 --warmup 1 --no-cpu-baseline; done; fi
if has rates; then
  WIDE="--field bn254_fr --rate 8 --rounds 8 57 --log2 18"
  ( timeout 300 python tools/sponge_rate.py $WIDE --absorb 11 --squeeze 9
    timeout 300 python tools/sponge_rate.py $WIDE --absorb 11 --squeeze 9 --mixed
    timeout 300 python tools/sponge_rate.py --rate 7 --log2 18 --absorb 10 --squeeze 8 --mixed
    timeout 300 python tools/sponge_rate.py --rate 4 --log2 19 --absorb 7 --squeeze 5 --mixed
    timeout 300 python tools/sponge_rate.py
    timeout 300 python tools/sponge_rate.py --mixed ) > $OUT/sponge_rate.txt 2>&1
  timeout 600 python tools/merkle_levels.py 21 > $OUT/merkle_levels.txt 2>&1
fi
if has hostpath; then
  ( timeout 600 python tools/host_path_rate.py 20; echo "--- the C ABI without the Python wrapper (tools/host_path_probe.cpp)"; timeout 300 ./tools/host_path_probe ) > $OUT/host_path.txt 2>&1
fi
if has rehearsal; then bash tools/gpu_group_rehearsal.sh $OUT/group_rehearsal > $OUT/group_rehearsal.txt 2>&1; fi
for f in sponge_rate merkle_levels host_path group_rehearsal; do [ -f $OUT/$f.txt ] && tail -30 $OUT/$f.txt; done
[ -f $OUT/pytest_gpu.log ] && tail -25 $OUT/pytest_gpu.log
[ -f $OUT/pytest_gpu_test_library.log ] && tail -5 $OUT/pytest_gpu_test_library.log
[ -f $OUT/smoke.log ] && tail -1 $OUT/smoke.log
for f in $OUT/bench_*.json; do [ -f $f ] && python - <<PY
import json
try:
    d=json.load(open("$f"))
    vi = d.get("valu_issue") or {}
    print("$f".split("/")[-1], "%.4g perm/s"%d["value"], "ms/step %.3f"%d["ms_per_step"], "mad frac %.3f (clk %.3g)"%(d["int_valu"]["frac"], d["int_valu"]["shader_clock_hz"]), "issue frac", vi.get("frac"), "engine", (d.get("engine") or {}).get("name"), "verified", d["verified"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$f", "unreadable:", e); print(open("$f".replace(".json",".err")).read()[-1500:])
PY
done
