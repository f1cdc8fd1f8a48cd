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


MIXED = "--mixed" in sys.argv    # per-sponge modes and positions drawn at random: lanes of a wave permute under EXEC masks
rng = np.random.default_rng(11)
h_state = synth.random_elements(field, n * 3, 10).view(np.int64).reshape(n, 3, 4).copy()
h_tag = rng.integers(0, 2, n).astype(np.int32)
h_idx = rng.integers(0, 3, n).astype(np.int32)

if MIXED:
    def fresh():      # noqa: F811
        return (torch.from_numpy(h_state).to(dev), torch.from_numpy(h_tag).to(dev), torch.from_numpy(h_idx).to(dev))


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
if MIXED:
    # permutations a sponge performs depend on its mode word; count them with the reference's rules (mod.rs:121-182, 232-254, 321-341)
    def perms_absorb(tag, idx, L, rate=2):
        k = np.where(tag == 0, idx, rate).astype(np.int64)        # Squeezing: permute first (treated as a full rate)
        return (k + L - 1) // rate, ((k + L - 1) % rate) + 1       # permutations, final next_absorb_index
    pa, idx_after = perms_absorb(h_tag, h_idx, 4)
    ps = 1 + (3 - 1) // 2                                          # after an absorb: Absorbing -> permute, squeeze 3 = 2 permutations
    print("mixed modes: absorb(4)  : %.3f ms  -> %.3e permutations/s (%.3f per sponge on average, max %d per wave)" % (a, pa.sum() / a * 1e3, pa.mean(), pa.max()))
    print("mixed modes: squeeze(3) : %.3f ms  -> %.3e permutations/s (%d per sponge)" % (s, ps * n / s * 1e3, ps))
else:
    print("absorb(4)  : %.3f ms  -> %.3e permutations/s (1 per sponge)" % (a, n / a * 1e3))
    print("squeeze(3) : %.3f ms  -> %.3e permutations/s (2 per sponge)" % (s, 2 * n / s * 1e3))
