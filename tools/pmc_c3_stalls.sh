#!/bin/bash
# Where the empty issue slots of the window kernels are: three SQ counter passes of one bench workload (default c3).
#   bash tools/pmc_c3_stalls.sh [workload] [out_dir]
W=${1:-c3}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=${2:-$R/gpurun_out/stalls_$W}
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
run() { timeout 600 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-verify > $OUT/$1.log 2>&1; python3 $R/tools/pmc_kernel_summary.py $OUT/$1 2>&1 | grep -A14 "permute_kernel" | head -16; }
run a "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"
run b "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM"
run c "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
