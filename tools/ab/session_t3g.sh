# round 5, session a: t = 3 on the window engine for every exponent (alpha = 17: the reference's default; 257: its weights table).
# parity on both sides of the threshold, then A/B old (round-4 HEAD: RegEngine<3,17,opt,tab>) vs new on the k3 workload, and today's c2 / c3 lines.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/device.txt
( timeout 1500 python -m pytest tests/test_gpu_sponge_passes.py tests/test_gpu_fullsize.py -x -q -m gpu -k "t3" ) > $O/pytest_t3.log 2>&1; tail -3 $O/pytest_t3.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or traces or merkle_trees or consistency" ) > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
WORKLOADS="k3 c2" STEPS=20 bash tools/ab/ab.sh 2>&1 | tee $O/ab_t3_every_exponent.txt
cp tools/ab/libposeidon_new.so sponge_amd/libposeidon_mi355x.so
timeout 600 python bench.py --workload k3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_k3.json 2> $O/bench_k3.err
timeout 600 python bench.py --workload c3 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python - <<'PY'
import json
for w in ("k3","c3"):
    try:
        d=json.load(open("gpurun_out/r05a/bench_%s.json"%w)); print(w, "%.4g"%d["value"], d["engine"]["name"], d["verified"], "clk %.3g"%d["int_valu"]["shader_clock_hz"])
    except Exception as e: print(w, "unreadable", e)
PY
