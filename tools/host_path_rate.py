#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point pmx_permute_batch (H2D + kernel + D2H, pageable host
memory) for the C2 batch.  Reported in DESIGN.md next to the device-resident rate; never bench.py's `value`."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sponge_amd as S
from sponge_amd import synth

f = S.BLS12_381_FR
cfg = S.poseidon_config_from_lfsr(f, 2, 5, 8, 31)
ctx = cfg.context(0)
n = 1 << 20
states = synth.random_elements(f, n * 3, 0x5EED0002).reshape(n, 3, 4)
ctx.permute_batch(states[:1024])
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    ctx.permute_batch(states)
    best = min(best, time.perf_counter() - t0)
out = {"pageable": {"entry": "pmx_permute_batch, pageable host buffer (includes the numpy copy of the wrapper)",
                    "states": n, "seconds": best, "permutations_per_s": n / best, "bytes_moved": 2 * n * 96}}
pin = S.pinned_empty((n, 3, 4))
pin[:] = states
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    ctx.permute_batch_inplace(pin)
    best = min(best, time.perf_counter() - t0)
out["pinned"] = {"entry": "pmx_permute_batch, page-locked host buffer (pmx_host_alloc): chunked H2D/kernel/D2H pipeline",
                 "states": n, "seconds": best, "permutations_per_s": n / best, "bytes_moved": 2 * n * 96}
print(json.dumps(out))
