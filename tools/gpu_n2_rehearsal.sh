#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a ONE-GPU box: two ranks share cuda:0, gloo instead of RCCL, gathers staged
# through the host.  Checks the sharding / gather / JSON plumbing only - the numbers are meaningless.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/n2
export MASTER_ADDR=127.0.0.1
run() {  # name, extra bench args
    local name=$1; shift
    PMX_BENCH_REHEARSAL=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) bench.py --gpus 2 --steps 5 --warmup 2 "$@" \
        > gpurun_out/n2/$name.log 2>&1
    echo "$name rc=$? $(grep -c '^{' gpurun_out/n2/$name.log) json line(s)"
    grep '^{' gpurun_out/n2/$name.log | python -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print("   value %.3e n_gpus %d units/step %s per-gpu %s scaling %s gather %s verified %s (%s)" % (d["value"], d["n_gpus"], d["config"]["permutations_per_step"], d["config"]["units_per_gpu"], d["scaling"], d["config"]["gather"], d["verified"], (d["verify"] or {}).get("what")))'
}
run c4_final --workload c2
run c2_step --workload c2 --total-log2 18 --gather step
run c2_none --workload c2 --total-log2 18 --gather none
run c2_ragged --workload c2 --total-units 100003
run c3 --workload c3
run c5 --workload c5 --total-log2 21
run h3 --workload h3
# and the real backend with both ranks on one device: RCCL is expected to refuse duplicate devices; record what it says
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29999 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/n2/nccl_same_gpu.log 2>&1
echo "nccl on one device rc=$?"; tail -3 gpurun_out/n2/nccl_same_gpu.log
