"""Throughput of the mid-stream sponge driver (pmx_sponge_absorb_batch_dev / pmx_sponge_squeeze_batch_dev) on 2^20
device-resident sponges: absorb 4 elements into fresh sponges (1 permutation each: the rate fills twice), then
squeeze 3 elements (2 permutations each).  BLS12-381 Fr, t = 3, alpha = 5."""
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
n = 1 << 20
inp = torch.from_numpy(synth.random_elements(field, n * 4, 9).view(np.int64).copy()).to(dev)
out = torch.zeros((n, 3, 4), dtype=torch.int64, device=dev)


def fresh():
    return (torch.zeros((n, 3, 4), dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev),
            torch.zeros(n, dtype=torch.int32, device=dev))


def run(reps):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t_abs = t_sq = 0.0
    for _ in range(reps):
        st, tag, idx = fresh()
        torch.cuda.synchronize()
        e0.record(stream)
        ctx.sponge_absorb_batch_dev(st.data_ptr(), tag.data_ptr(), idx.data_ptr(), inp.data_ptr(), 4, n, stream.cuda_stream)
        e1.record(stream)
        ctx.sponge_squeeze_batch_dev(st.data_ptr(), tag.data_ptr(), idx.data_ptr(), out.data_ptr(), 3, n, stream.cuda_stream)
        e2.record(stream)
        torch.cuda.synchronize()
        t_abs += e0.elapsed_time(e1)
        t_sq += e1.elapsed_time(e2)
    return t_abs / reps, t_sq / reps


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    run(1)
a, s = run(10)
print("absorb(4)  : %.3f ms  -> %.3e permutations/s (1 per sponge)" % (a, n / a * 1e3))
print("squeeze(3) : %.3f ms  -> %.3e permutations/s (2 per sponge)" % (s, 2 * n / s * 1e3))
