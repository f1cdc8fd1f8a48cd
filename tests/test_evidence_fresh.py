"""The counter figures bench.py quotes (profiles/valu_instructions.json: SQ_INSTS_VALU per permutation; profiles/hbm_traffic.json:
FETCH_SIZE / WRITE_SIZE per launch) are committed constants, taken by rocprofv3 --pmc passes that cannot run inside a timed region.  They
belong to the device code they were taken on: both files carry tools/source_hash.py's hash of sponge_amd/csrc/*.hip, *.hpp and the
Makefile.  This test fails when the tree's kernels have moved on from that hash - re-run the counter passes (tools/gpu_r06.sh: stages
slots pmc, then tools/install_evidence.sh) - and bench.py, which makes the same comparison, says "stale" instead of quoting them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from source_hash import kernel_source_hash  # noqa: E402


def test_committed_counter_figures_were_taken_on_this_trees_kernels():
    want = kernel_source_hash()
    for name in ("valu_instructions.json", "hbm_traffic.json"):
        recs = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert recs.get("_kernel_source_hash") == want, (
            "profiles/%s was taken on kernel sources %s, this tree is %s: re-run the counter passes" % (name, recs.get("_kernel_source_hash"), want))
        assert any(isinstance(v, dict) for v in recs.values())


def test_bench_refuses_to_quote_stale_figures(tmp_path, monkeypatch):
    """bench.py's own check: with another hash in the file the loaders return "stale", not a number."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    assert bench.evidence_stale({"_kernel_source_hash": kernel_source_hash()}) is None
    msg = bench.evidence_stale({"_kernel_source_hash": "0123456789abcdef"})
    assert msg and msg.startswith("stale")
    assert bench.evidence_stale({}).startswith("stale")
