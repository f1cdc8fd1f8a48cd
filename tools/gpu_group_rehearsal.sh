#!/bin/bash
# bench.py's N > 1 code path on a ONE-GPU box THROUGH THE PRODUCT'S DEVICE GROUP (PMX_BENCH_REHEARSAL=group): W ranks under
# torch.distributed.run share cuda:0, torch.distributed's control plane runs on gloo, and every data-path call is the one
# a real multi-GPU run makes - pmx_mgpu_create_rank, pmx_mgpu_permute_shards_dev, pmx_mgpu_all_gather_dev (equal and ragged),
# pmx_mgpu_gather_dev, pmx_mgpu_permute_gather_dev, pmx_mgpu_merkle_2to1_dev - with the collective library named by PMX_RCCL_LIBRARY: the tests' stand-in (tests/fake_rccl, ranks
# in different processes; RCCL itself refuses two ranks on one device).  Every rank verifies its whole gathered copy.
# (PMX_BENCH_REHEARSAL=group makes bench.py bind the test-hook build of the library, libposeidon_mi355x_test.so: only that build
# reads PMX_RCCL_LIBRARY.)  The numbers are meaningless; what counts is rc = 0, "verified": true and rccl.ranks = W on every line.
# Launcher-free forms (round 5): `python bench.py --gpus W` starts its own ranks as a child process; `--single-process` drives W
# device slots from one process (pmx_mgpu_create).
#   bash tools/gpu_group_rehearsal.sh [out_dir]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/group_rehearsal}
mkdir -p "$OUT"
make -s -C tests/fake_rccl all
export MASTER_ADDR=127.0.0.1 PMX_BENCH_REHEARSAL=group FAKE_RCCL_XPROC=1
export PMX_RCCL_LIBRARY="$PWD/tests/fake_rccl/librccl.so.1"
fails=0
run() {  # name, world, extra bench args
    local name=$1 w=$2; shift 2
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$w" --master-addr 127.0.0.1 \
        --master-port $((29500 + RANDOM % 500)) bench.py --gpus "$w" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    grep '^{' "$OUT/$name.log" > "$OUT/$name.json"
    python - "$OUT/$name.json" "$name" "$w" "$rc" <<'PY' || fails=$((fails + 1))
import json, sys
path, name, w, rc = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
lines = [json.loads(l) for l in open(path)]
ok = rc == 0 and len(lines) == 1 and lines[0]["verified"] is True and lines[0]["n_gpus"] == w and (lines[0].get("rccl") or {}).get("ranks") == w
d = lines[0] if lines else {}
print("%-14s W=%d rc=%d %s  units/step %s per-gpu %s gather %s rccl.ranks %s (%s)" % (
    name, w, rc, "OK  " if ok else "FAIL", d.get("config", {}).get("permutations_per_step"), d.get("config", {}).get("units_per_gpu"),
    d.get("config", {}).get("gather"), (d.get("rccl") or {}).get("ranks"), (d.get("verify") or {}).get("what")))
sys.exit(0 if ok else 1)
PY
}
run c2_w2_final 2 --workload c2 --total-log2 18
run c2_w2_step 2 --workload c2 --total-log2 16 --gather step
run c2_w3_ragged 3 --workload c2 --total-units 100003
run c2_w8_final 8 --workload c2 --total-log2 18
run c2_w8_ragged 8 --workload c2 --total-units 100003 --gather step
run c2_w3_root 3 --workload c2 --total-units 100003 --gather root
run c2_w8_overlap 8 --workload c2 --total-log2 18 --gather overlap
run c2_w3_overlap_root 3 --workload c2 --total-units 100003 --gather overlap-root --gather-chunks 5
run c3_w2 2 --workload c3 --total-log2 14
run c5_w2 2 --workload c5 --total-log2 16
run c5_w8 8 --workload c5 --total-log2 18
run h3_w2 2 --workload h3 --total-log2 14
# the same without a launcher around bench.py: it starts the ranks itself / drives every slot from one process
bare() {  # name, world, extra bench args
    local name=$1 w=$2; shift 2
    ( unset WORLD_SIZE RANK LOCAL_RANK MASTER_PORT; timeout 900 python bench.py --gpus "$w" --steps 3 --warmup 1 "$@" > "$OUT/$name.log" 2> "$OUT/$name.err" )
    local rc=$?
    grep '^{' "$OUT/$name.log" > "$OUT/$name.json"
    python - "$OUT/$name.json" "$name" "$w" "$rc" <<'PY' || fails=$((fails + 1))
import json, sys
path, name, w, rc = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
lines = [json.loads(l) for l in open(path)]
ok = rc == 0 and len(lines) == 1 and lines[0]["verified"] is True and lines[0]["n_gpus"] == w and (lines[0].get("rccl") or {}).get("ranks") == w
d = lines[0] if lines else {}
print("%-22s W=%d rc=%d %s  via %s" % (name, w, rc, "OK  " if ok else "FAIL", ((d.get("rccl") or {}).get("via") or "")[:60]))
sys.exit(0 if ok else 1)
PY
}
bare bare_c2_w2 2 --workload c2 --total-log2 18
bare bare_c2_w8 8 --workload c2 --total-units 100003
bare bare_c5_w8 8 --workload c5 --total-log2 18
( unset FAKE_RCCL_XPROC
bare single_c2_w2 2 --single-process --workload c2 --total-units 100003
bare single_c2_w8 8 --single-process --workload c2 --total-log2 18
bare single_c2_w3_root 3 --single-process --workload c2 --total-units 100003 --gather root
bare single_c2_w8_overlap 8 --single-process --workload c2 --total-log2 18 --gather overlap
bare single_c2_w3_overlap_root 3 --single-process --workload c2 --total-units 100003 --gather overlap-root --gather-chunks 3
bare single_c5_w8 8 --single-process --workload c5 --total-log2 18 )
ls /dev/shm | grep -c '^fake_rccl_' | sed 's/^/leftover shared-memory objects: /'
echo "failures: $fails"
exit $fails
