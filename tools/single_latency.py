"""Latency of small batches through the device-resident entry points (n = 1 ... 2^18 states, t = 3)."""
import os
import sys
import time

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
nmax = 1 << 18
buf = torch.from_numpy(synth.random_elements(field, nmax * 3, 5).view(np.int64).copy()).to(dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    ctx.permute_batch_dev(buf.data_ptr(), nmax, stream.cuda_stream)
    torch.cuda.synchronize()
for k in (0, 4, 6, 8, 10, 12, 14, 15, 16, 17, 18):
    n = 1 << k
    for _ in range(3):
        ctx.permute_batch_dev(buf.data_ptr(), n, stream.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(30):
        ctx.permute_batch_dev(buf.data_ptr(), n, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print("permute n=2^%-2d  %8.4f ms  %.3e perm/s" % (k, ms, n / ms * 1e3))
