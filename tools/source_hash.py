#!/usr/bin/env python3
"""A hash of everything that decides what the device code is: sponge_amd/csrc/*.hip, *.hpp and the Makefile (its flags).  The counter
passes of a round write it next to their figures (profiles/valu_instructions.json, profiles/hbm_traffic.json: key "_kernel_source_hash");
bench.py quotes those committed figures only while the hash still matches (otherwise the line says "stale"), and
tests/test_evidence_fresh.py fails when the tree's kernels have moved on from the ones the figures were taken on.
usage: source_hash.py          prints the hash of this tree"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash(root: str = ROOT) -> str:
    csrc = os.path.join(root, "sponge_amd", "csrc")
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) + [os.path.join(csrc, "Makefile")]):
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
