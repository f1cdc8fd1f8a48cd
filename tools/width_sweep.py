#!/usr/bin/env python3
"""Device-resident permutation throughput for every width of the reference's default BLS12-381 table (and a few
configs that land on the fallback engine).  Prints one JSON object; numbers go into DESIGN.md."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sponge_amd as S
from sponge_amd import synth

f = S.BLS12_381_FR
rows = []
cases = [("default constraints rate %d" % r, S.get_default_poseidon_parameters(f, r, False)) for r in range(2, 9)]
cases += [("default weights rate 2 (alpha 257)", S.get_default_poseidon_parameters(f, 2, True)),
          ("default weights rate 8 (alpha 257)", S.get_default_poseidon_parameters(f, 8, True)),
          ("t=5 alpha=17 (generic S-box hybrid)", S.poseidon_config_from_lfsr(f, 4, 17, 8, 56))]
stream = torch.cuda.current_stream()
for name, cfg in cases:
    t = cfg.t
    n = 1 << (20 if t <= 4 else 18)
    host = synth.random_elements(f, n * t, 1)
    d = torch.from_numpy(host.view(np.int64).copy()).cuda()
    ctx = cfg.context(0)
    for _ in range(2):
        ctx.permute_batch_dev(d.data_ptr(), n, stream.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ctx.permute_batch_dev(d.data_ptr(), n, stream.cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    rows.append({"config": name, "t": t, "alpha": cfg.alpha, "rounds": [cfg.full_rounds, cfg.partial_rounds], "states": n,
                 "ms": round(ms, 3), "permutations_per_s": n / ms * 1e3})
print(json.dumps(rows))
