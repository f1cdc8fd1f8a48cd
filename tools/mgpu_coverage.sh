#!/bin/bash
# Line coverage of sponge_amd/csrc/pmx_mgpu.cpp (host code) under the stand-in tests, on a ONE-GPU box:
#   tools/mgpu_coverage.sh [OUT_DIR]        (default gpurun_out/mgpu_cov)
# builds tests/cov/libposeidon_mi355x_cov.so (the product's objects + a gcov-instrumented pmx_mgpu.o), runs
# tests/mgpu_standin_worker.py for W = 2, 3, 8 behind tests/fake_rccl, once behind the library that lacks a symbol, once with
# the library named by PMX_RCCL_LIBRARY and once with a name that cannot be loaded,
# then gcov.  Writes pmx_mgpu.cpp.gcov (annotated source) and coverage_summary.txt.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/mgpu_cov}"
mkdir -p "$OUT"
OUT="$(cd "$OUT" && pwd)"
cd "$ROOT"
# PMX_COV_NO_BUILD=1: use the libraries that travelled with the snapshot (sponge_amd/csrc/build/ does not travel to the
# GPU box, and rebuilding the three device translation units there costs GPU-minutes for nothing)
if [ "${PMX_COV_NO_BUILD:-0}" != "1" ]; then
    make -C sponge_amd/csrc -j4 all >/dev/null && make -C oracle all >/dev/null && make -C tests/fake_rccl all >/dev/null || exit 1
    make -C tests/cov all >/dev/null || exit 1
fi
rm -f tests/cov/*.gcda
COV="$ROOT/tests/cov/libposeidon_mi355x_cov.so"
status=0
for W in 2 3 8; do
    LD_LIBRARY_PATH="$ROOT/tests/fake_rccl:${LD_LIBRARY_PATH:-}" python3 tests/mgpu_standin_worker.py $W "$OUT/standin_w$W.json" "$COV" \
        > "$OUT/standin_w$W.log" 2>&1 || status=1
done
# the "librccl has no symbol" path of the loader
LD_LIBRARY_PATH="$ROOT/tests/fake_rccl/broken:${LD_LIBRARY_PATH:-}" python3 - "$COV" > "$OUT/broken.log" 2>&1 <<'PY' || status=1
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from sponge_amd import _lib
_lib.use_library(sys.argv[1], test_hooks=True)
import sponge_amd as S
from sponge_amd import mgpu
from gpu_helpers import product_config
try:
    mgpu.DeviceGroup.single_process(product_config("bls_t3_a5_8_31"), 1)
    sys.exit(2)
except S.PmxError as e:
    assert "librccl has no symbol ncclBroadcast" in str(e), str(e)
    print("ok:", e)
PY
# PMX_RCCL_LIBRARY: a named library that loads (the stand-in, by path instead of by search order) and one that does not
PMX_RCCL_LIBRARY="$ROOT/tests/fake_rccl/librccl.so.1" python3 tests/mgpu_standin_worker.py 2 "$OUT/standin_named_w2.json" "$COV" \
    > "$OUT/standin_named_w2.log" 2>&1 || status=1
PMX_RCCL_LIBRARY=/nonexistent/librccl.so python3 - "$COV" > "$OUT/unloadable.log" 2>&1 <<'PY' || status=1
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from sponge_amd import _lib
_lib.use_library(sys.argv[1], test_hooks=True)
import sponge_amd as S
from sponge_amd import mgpu
try:
    mgpu.unique_id()
    sys.exit(2)
except S.PmxError as e:
    assert "/nonexistent/librccl.so could not be loaded" in str(e), str(e)
    print("ok:", e)
PY
(cd tests/cov && gcov -o . pmx_mgpu_cov.o > "$OUT/gcov_stdout.txt" 2>&1)
cp tests/cov/pmx_mgpu.cpp.gcov "$OUT/" 2>/dev/null || status=1
python3 - "$OUT" <<'PY'
import re, sys, json, os
out = sys.argv[1]
hit = miss = 0
missed = []
for line in open(os.path.join(out, "pmx_mgpu.cpp.gcov"), errors="replace"):
    m = re.match(r"\s*([^:]+):\s*(\d+):(.*)", line)
    if not m:
        continue
    count, no, text = m.group(1).strip(), int(m.group(2)), m.group(3)
    if no == 0 or count == "-":
        continue
    if count in ("#####", "====="):
        miss += 1
        missed.append(f"{no}: {text.strip()}")
    else:
        hit += 1
lines = [f"pmx_mgpu.cpp: {hit} of {hit + miss} executable lines hit = {100.0 * hit / max(hit + miss, 1):.1f} %", "", "lines never executed:"] + missed
lines += ["", "scenarios:"]
for w in (2, 3, 8):
    try:
        r = json.load(open(os.path.join(out, f"standin_w{w}.json")))
        bad = [s["name"] for s in r["scenarios"] if not s["ok"]]
        lines.append(f"  W = {w}: {len(r['scenarios'])} scenarios, failed: {bad or 'none'}; library {os.path.basename(r.get('library', '?'))}; stand-in stats {r.get('fake_rccl_stats')}")
    except Exception as e:
        lines.append(f"  W = {w}: no result ({e})")
open(os.path.join(out, "coverage_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:3]))
PY
exit $status
