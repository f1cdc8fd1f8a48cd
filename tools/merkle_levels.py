"""Time of a 2-to-1 Merkle reduction as a function of the leaf count (device-resident), and the per-level cost
derived from it: level with m parents ~= T(2m leaves) - T(m leaves).  Run on the GPU box."""
import json
import sys
import time

import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sponge_amd as S  # noqa: E402
from sponge_amd import synth  # noqa: E402

field = S.FIELDS["bls12_381_fr"]
cfg = S.poseidon_config_from_lfsr(field, 2, 5, 8, 31)
ctx = cfg.context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
max_log = int(sys.argv[1]) if len(sys.argv) > 1 else 21
host = synth.random_elements(field, 1 << max_log, 77)
rows = []
prev = 0.0
# spin-up
nodes = torch.zeros((2 * (1 << max_log) - 1, 4), dtype=torch.int64, device=dev)
nodes[:1 << max_log] = torch.from_numpy(host.view(np.int64).copy()).to(dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    ctx.merkle_2to1_dev(nodes.data_ptr(), 1 << max_log, stream.cuda_stream)
    torch.cuda.synchronize()
for k in range(1, max_log + 1):
    n = 1 << k
    reps = 20 if k > 16 else 50
    for _ in range(3):
        ctx.merkle_2to1_dev(nodes.data_ptr(), n, stream.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        ctx.merkle_2to1_dev(nodes.data_ptr(), n, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    rows.append({"leaves_log2": k, "tree_ms": ms, "top_level_ms": ms - prev, "parents": n // 2,
                 "level_perm_per_s": (n // 2) / ((ms - prev) * 1e-3) if ms > prev else None})
    prev = ms
for r in rows:
    print("2^%-2d leaves  tree %8.4f ms   level(%7d parents) %8.4f ms  %s" % (
        r["leaves_log2"], r["tree_ms"], r["parents"], r["top_level_ms"],
        ("%.3e perm/s" % r["level_perm_per_s"]) if r["level_perm_per_s"] else ""))
print(json.dumps(rows))
