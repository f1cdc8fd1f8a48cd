#!/bin/bash
# Instruction-fetch, scalar-memory and LDS levels of one bench workload's kernels (what a wave waits for besides the VALU port).
#   bash tools/pmc_fetch_levels.sh [workload] [out_dir]
W=${1:-c3}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=${2:-$R/gpurun_out/levels_$W}
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
run() { timeout 600 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/$1.log 2>&1; python3 $R/tools/pmc_kernel_summary.py $OUT/$1 2>&1 | grep -A10 "permute_kernel" | head -12; }
run a "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_BUSY_CYCLES"
run b "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"
