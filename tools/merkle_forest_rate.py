"""2^21 leaves as ONE 2-to-1 tree and as a forest of smaller trees advanced together (pmx_merkle_2to1_forest_dev): the narrow top
levels of one tree are latency-bound, a level of the forest is n_trees times as wide.  Device-resident; run on the GPU box.
usage: python tools/merkle_forest_rate.py [log2 of the total leaves, default 21]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sponge_amd as S  # noqa: E402
from sponge_amd import synth  # noqa: E402

total_log = int(sys.argv[1]) if len(sys.argv) > 1 else 21
field = S.FIELDS["bls12_381_fr"]
cfg = S.poseidon_config_from_lfsr(field, 2, 5, 8, 31)
ctx = cfg.context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
total = 1 << total_log
nodes = torch.zeros((2 * total, 4), dtype=torch.int64, device=dev)
nodes[:total] = torch.from_numpy(synth.random_elements(field, total, 77).view(np.int64).copy()).to(dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    ctx.merkle_2to1_dev(nodes.data_ptr(), total, stream.cuda_stream)
    torch.cuda.synchronize()
for trees_log in (0, 3, 5, 8, 11, 14, 16, total_log - 1):
    n_trees, m = 1 << trees_log, 1 << (total_log - trees_log)
    reps = 20
    for _ in range(3):
        ctx.merkle_2to1_forest_dev(nodes.data_ptr(), n_trees, m, stream.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        ctx.merkle_2to1_forest_dev(nodes.data_ptr(), n_trees, m, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    comps = total - n_trees
    print("2^%-2d trees of 2^%-2d leaves: %8.3f ms  %9d compressions  %.3e /s  (%d launches)" % (
        trees_log, total_log - trees_log, ms, comps, comps / ms * 1e3, total_log - trees_log))
