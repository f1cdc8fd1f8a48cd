#!/bin/bash
# The first visit to a multi-GPU node, as ONE command:  bash tools/first_8gpu_run.sh [max_gpus] [out_dir]
#   1. tests/test_gpu_mgpu.py on real RCCL with every GPU of the box (device groups of 1..N, ragged gathers, the sharded tree, one process
#      per GPU with pmx_mgpu_create_rank) - every rank checked against the C restatement;
#   2. BASELINE configs[1] / [3] (c2: 2^20 states at N = 1, the 2^24-state batch sharded at N = 2, 4, 8, one final RCCL gather inside the
#      timed region; at N > 1 also --gather root / overlap / overlap-root) and configs[4] (c5: the 2^24-leaf tree) at N = 1, 2, 4, 8, each in BOTH launcher-free forms of bench.py:
#      `python bench.py --gpus N` (starts one rank per GPU as a child `python -m torch.distributed.run`) and `--single-process` (one process,
#      pmx_mgpu_create = ncclCommInitAll);
#   3. a table: permutations/s, ms per step, gather_ms next to the 1.3 ms DESIGN.md section 6 predicts for the C4 gather (192 MiB per peer
#      over its own xGMI link), rccl.ranks, verified.
# Nothing here needs a launcher or an environment variable; every line must have rc = 0, verified = true, rccl.ranks = N.
set -u
cd "$(dirname "$0")/.."
MAX=${1:-8}
OUT=${2:-scale_out}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
unset WORLD_SIZE RANK LOCAL_RANK MASTER_PORT
PY=${PYTHON:-python}
have=$($PY -c "from sponge_amd import _lib; print(_lib.lib().pmx_device_count())" 2> "$OUT/device_count.err")
case "$have" in
  ''|*[!0-9]*) echo "cannot count the devices (library missing or not importable: $(tail -1 "$OUT/device_count.err")); nothing run" | tee "$OUT/summary.txt"; exit 2;;
esac
[ "$have" -ge 1 ] || { echo "no HIP device visible; nothing run" | tee "$OUT/summary.txt"; exit 2; }
[ "$have" -lt "$MAX" ] && MAX=$have
echo "devices visible: $have, running up to N = $MAX" | tee "$OUT/summary.txt"
( timeout 3000 $PY -m pytest tests/test_gpu_mgpu.py -x -q -m gpu ) > "$OUT/pytest_mgpu.log" 2>&1
echo "tests/test_gpu_mgpu.py: exit $? - $(tail -1 "$OUT/pytest_mgpu.log")" | tee -a "$OUT/summary.txt"
for n in 1 2 4 8; do
  [ "$n" -le "$MAX" ] || break
  for w in c2 c5; do
    steps=20; [ "$w" = c5 ] && steps=5
    timeout 1800 $PY bench.py --gpus $n --steps $steps --warmup 5 --workload $w --no-cpu-baseline > "$OUT/${w}_n${n}_ranks.json" 2> "$OUT/${w}_n${n}_ranks.err"
    echo "rc $? ${w} N=$n ranks" >> "$OUT/rc.txt"
    if [ "$n" -gt 1 ]; then
      timeout 1800 $PY bench.py --gpus $n --single-process --steps $steps --warmup 5 --workload $w > "$OUT/${w}_n${n}_single.json" 2> "$OUT/${w}_n${n}_single.err"
      echo "rc $? ${w} N=$n single" >> "$OUT/rc.txt"
      if [ "$w" = c2 ]; then   # the other forms of the epilogue: to rank 0 only, and piece by piece behind the last step's kernels
        for g in root overlap overlap-root; do
          timeout 1800 $PY bench.py --gpus $n --steps $steps --warmup 5 --workload c2 --gather $g --no-cpu-baseline > "$OUT/c2_n${n}_ranks_${g}.json" 2> "$OUT/c2_n${n}_ranks_${g}.err"
          echo "rc $? c2 N=$n ranks --gather $g" >> "$OUT/rc.txt"
        done
      fi
    fi
  done
done
$PY - "$OUT" <<'PY' | tee -a "$OUT/summary.txt"
import glob, json, os, sys
out = sys.argv[1]
print(open(os.path.join(out, "rc.txt")).read())
print("%-32s %3s %12s %10s %10s %10s %6s %s" % ("run", "N", "perm/s", "ms/step", "gather_ms", "predicted", "ranks", "verified"))
for f in sorted(glob.glob(os.path.join(out, "c?_n*_*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print("%-32s %3d %12.4g %10.3f %10s %10s %6s %s" % (os.path.basename(f)[:-5], d["n_gpus"], d["value"], d["ms_per_step"],
              "%.3f" % d["gather_ms"] if d.get("gather_ms") is not None else "-",
              "%.3f" % d["gather_ms_predicted"] if d.get("gather_ms_predicted") is not None else "-", (d.get("rccl") or {}).get("ranks"), d["verified"]))
    except Exception as e:
        print("%-32s no JSON line (%s): %s" % (os.path.basename(f), e, open(f[:-5] + ".err").read()[-300:].replace("\n", " | ")))
print("(predicted: one shard over one xGMI link at 153 GB/s - ~1.3 ms for the C4 gather at N = 8, to every rank or to rank 0 alike; the overlap forms")
print(" leave one piece's transfer behind the last kernel.  The overlap forms have no epilogue to time (gather_ms '-'): compare their ms/step x steps with the other forms'.)")
PY
