#!/bin/bash
# The scaling series of BASELINE.md section 2 on a multi-GPU node: C2 at N = 1, C4 (2^24 states, RCCL gather) and C5
# (2^24-leaf tree) at N = 2, 4, 8.  One JSON line per run into scale_out/.  Usage: bash tools/run_scale.sh [max_gpus]
set -u
cd "$(dirname "$0")/.."
MAX=${1:-8}
OUT=scale_out
mkdir -p $OUT
export MASTER_ADDR=127.0.0.1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/c2_n1.json 2> $OUT/c2_n1.err
python bench.py --gpus 1 --steps 5 --warmup 2 --workload c5 --no-cpu-baseline > $OUT/c5_n1.json 2> $OUT/c5_n1.err
for n in 2 4 8; do
  [ $n -le $MAX ] || break
  for w in c2 c5; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
      bench.py --gpus $n --steps 20 --warmup 5 --workload $w > $OUT/${w}_n$n.json 2> $OUT/${w}_n$n.err
  done
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("scale_out/*.json")):
    try:
        line = [l for l in open(f) if l.startswith("{")][-1]
        d = json.loads(line)
        print(f, "n_gpus", d["n_gpus"], "%.4g perm/s" % d["value"], "ms/step %.3f" % d["ms_per_step"], "verified", d["verified"],
              "rccl", (d.get("rccl") or {}).get("ranks"), "gather_ms", d.get("gather_ms"))
    except Exception as e:
        print(f, "no JSON line:", e)
PY
