"""Throughput of the mid-stream sponge driver (pmx_sponge_absorb_batch_dev / pmx_sponge_squeeze_batch_dev) on device-resident
sponges with explicit (state, mode) words - the absorb / squeeze batch driver of src/poseidon/mod.rs:121-182, 232-254, 321-341.

    python tools/sponge_rate.py [--field bls12_381_fr|bn254_fr] [--rate R] [--rounds RF RP] [--log2 N] [--absorb L] [--squeeze K] [--mixed]

Defaults: BLS12-381 Fr, rate 2 (t = 3), 8 + 31 rounds, 2^20 sponges, absorb(4) into fresh sponges then squeeze(3).
`--field bn254_fr --rate 8 --rounds 8 57 --log2 18 --absorb 11 --squeeze 9` is the wide-state driver of BASELINE's configs[2].
--mixed: per-sponge mode words drawn at random (Absorbing / Squeezing, index in [0, rate]).  The permutations a sponge
performs are counted with the reference's rules, so the rate printed is permutations the REFERENCE would execute per second."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sponge_amd as S  # noqa: E402
from sponge_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--field", default="bls12_381_fr")
ap.add_argument("--rate", type=int, default=2)
ap.add_argument("--alpha", type=int, default=5)
ap.add_argument("--rounds", type=int, nargs=2, default=None)
ap.add_argument("--log2", type=int, default=None)
ap.add_argument("--absorb", type=int, default=None)
ap.add_argument("--squeeze", type=int, default=None)
ap.add_argument("--mixed", action="store_true")
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()

field = S.FIELDS[args.field]
rate = args.rate
t = rate + 1
rf, rp = args.rounds if args.rounds else ((8, 31) if t == 3 else (8, 57) if t >= 6 else (8, 56))
cfg = S.poseidon_config_from_lfsr(field, rate, args.alpha, rf, rp)
ctx = cfg.context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
n = 1 << (args.log2 if args.log2 is not None else (20 if t == 3 else 18))
L = args.absorb if args.absorb is not None else (4 if t == 3 else rate + 3)
K = args.squeeze if args.squeeze is not None else (3 if t == 3 else rate + 1)
inp = torch.from_numpy(synth.random_elements(field, n * max(L, 1), 9).view(np.int64).copy()).to(dev)
out = torch.zeros((n, max(K, 1), 4), dtype=torch.int64, device=dev)

rng = np.random.default_rng(11)
h_state = synth.random_elements(field, n * t, 10).view(np.int64).reshape(n, t, 4).copy()
if args.mixed:
    h_tag = rng.integers(0, 2, n).astype(np.int32)
    h_idx = rng.integers(0, rate + 1, n).astype(np.int32)
else:
    h_state[:] = 0                                       # CryptographicSponge::new
    h_tag = np.zeros(n, dtype=np.int32)
    h_idx = np.zeros(n, dtype=np.int32)


# The initial sponges live on the device and every repetition starts from device-to-device copies of them: an upload from
# host memory in front of each timed call idles the shader engines for milliseconds, the clock drops, and the call is then
# timed on the ramp (measured: t = 3 absorb(4) 1.55 ms on a busy device, 1.70-1.74 ms behind a 96 MB upload).
d_state0, d_tag0, d_idx0 = torch.from_numpy(h_state).to(dev), torch.from_numpy(h_tag).to(dev), torch.from_numpy(h_idx).to(dev)


def fresh():
    return d_state0.clone(), d_tag0.clone(), d_idx0.clone()


def perms_absorb(tag, idx, length):
    """permutations of absorb(length) per sponge and the index afterwards (mod.rs:232-254, 121-150)"""
    if length == 0:
        return np.zeros(len(tag), dtype=np.int64), idx.astype(np.int64), tag
    k = np.where(tag == 0, idx, rate).astype(np.int64)             # Squeezing: permute first (like a full rate)
    return (k + length - 1) // rate, ((k + length - 1) % rate) + 1, np.zeros_like(tag)


def perms_squeeze(tag, idx, length):
    """permutations of squeeze_native_field_elements(length) per sponge (mod.rs:321-341, 153-182 incl. the :175 test)"""
    idx = idx.astype(np.int64)
    first = (tag == 0) | (idx == rate)                              # Absorbing: always; Squeezing: iff the rate is used up
    i0 = np.where(first, 0, idx)
    fits = i0 + length <= rate
    rem1 = length - (rate - i0)                                     # left after the first (partial) chunk
    more = np.where(fits, 0, np.where(length != rate, 1, 0) + np.maximum(0, -(-rem1 // rate) - 1))
    return first.astype(np.int64) + more


def run(reps):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t_abs = t_sq = 0.0
    for _ in range(reps):
        st, tag, idx = fresh()      # (same stream as the calls below: ordered, no host synchronisation in between)
        e0.record(stream)
        ctx.sponge_absorb_batch_dev(st.data_ptr(), tag.data_ptr(), idx.data_ptr(), inp.data_ptr(), L, n, stream.cuda_stream)
        e1.record(stream)
        ctx.sponge_squeeze_batch_dev(st.data_ptr(), tag.data_ptr(), idx.data_ptr(), out.data_ptr(), K, n, stream.cuda_stream)
        e2.record(stream)
        torch.cuda.synchronize()
        t_abs += e0.elapsed_time(e1)
        t_sq += e1.elapsed_time(e2)
    return t_abs / reps, t_sq / reps


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    run(1)
a, s = run(args.reps)
pa, idx_after, tag_after = perms_absorb(h_tag, h_idx, L)
ps = perms_squeeze(tag_after, idx_after, K)
label = "%s t=%d %d+%d, 2^%d sponges, %s" % (args.field, t, rf, rp, n.bit_length() - 1, "mixed modes" if args.mixed else "fresh sponges")
for name, ms, p in (("absorb(%d)" % L, a, pa), ("squeeze(%d)" % K, s, ps)):
    tot = int(p.sum())
    print("%s: %-11s: %8.3f ms -> %.3e permutations/s (%.3f per sponge on average, max %d)" % (
        label, name, ms, (tot / ms * 1e3) if ms > 0 else 0.0, p.mean(), int(p.max())))
