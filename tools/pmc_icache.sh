#!/bin/bash
# Instruction-cache counters of one bench workload's kernels (the t = 9 kernel is ~73 KB of code, its window loop 51 KB; the cache holds 64 KB
# and two CUs share it).   bash tools/pmc_icache.sh [workload] [out_dir]
W=${1:-c3}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=${2:-$R/gpurun_out/icache_$W}
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/available.txt; echo >> $OUT/available.txt
cat $OUT/available.txt
run() { timeout 600 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/$1.log 2>&1; python3 $R/tools/pmc_kernel_summary.py $OUT/$1 2>&1 | grep -A10 "permute_kernel" | head -12; }
run a "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
run b "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES"
