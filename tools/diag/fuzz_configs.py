#!/usr/bin/env python3
"""Seeded random Poseidon configurations on the GPU against the C port (oracle/): every exponent class (0, 1, small, the usual, 64-bit),
odd and zero round counts, every rate / capacity split of widths 2 ... 12, both benchmarked fields and (every fourth config) a random prime of 225 ... 255 bits; per config whole permutations at several batch
sizes (both sides of the engine thresholds at t = 3), the fixed-shape hash, a small tree and (every other config) the duplex driver on sponges in mixed modes.  Prints one line per failing case and a
summary; exit code 1 if anything differs.      usage: tools/diag/fuzz_configs.py [n_configs] [seed]"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
import numpy as np
import sponge_amd as S
from sponge_amd import _lib, synth
from oracle import cref, poseidon_oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
FIELDS = {"bls12_381_fr": (O.BLS12_381_FR, 255), "bn254_fr": (O.BN254_FR, 254)}


def _is_prime(n):
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def random_prime(bits):
    """a prime of exactly `bits` bits (the library takes 225 ... 255: pmx_prepare.hpp)"""
    while True:
        c = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        if _is_prime(c):
            return c
ALPHAS = [0, 1, 1, 2, 3, 4, 5, 5, 5, 6, 7, 11, 17, 17, 257, 65537, (1 << 32) + 1, (1 << 63) + 1, (1 << 64) - 1]
bad = 0
t0 = time.time()
for k in range(N):
    if k % 4 == 3:                       # every fourth config over a random prime: any size the library takes, any top byte
        bits = rng.choice([225, 226, 233, 240, 247, 248, 249, 253, 254, 255])
        p = random_prime(bits)
        fname = "prime%d_%x" % (bits, p >> (bits - 16))
        f = S.Field(fname, p)
    else:
        fname = rng.choice(list(FIELDS))
        p, bits = FIELDS[fname]
        f = S.FIELDS[fname]
    t = rng.choice([2, 3, 3, 3, 4, 5, 6, 7, 8, 9, 9, 10, 12])
    capacity = rng.choice([1, 1, 1, 0, 2, 3])
    capacity = min(capacity, t - 1)
    rate = t - capacity
    alpha = rng.choice(ALPHAS)
    rf = rng.choice([0, 1, 2, 3, 4, 6, 7, 8, 8, 8, 10])
    rp = rng.choice([0, 1, 2, 3, 5, 6, 7, 13, 22, 31, 56, 57, 60, 66, 67, 70])
    if rf + rp == 0:
        rf = 2
    base = S.poseidon_config_from_lfsr(f, t - 1, alpha, rf, rp)
    cfg = S.PoseidonConfig(f, rf, rp, alpha, base.mds, base.ark, rate, capacity)
    ob = O.make_config(p, bits, t - 1, alpha, rf, rp)
    cr = cref.CRef(O.PoseidonConfig(ob.p, rf, rp, alpha, ob.ark, ob.mds, rate, capacity))
    ctx = cfg.context()
    info = _lib.PmxEngineInfo()
    what = "%s t=%d rate=%d cap=%d alpha=%d rf=%d rp=%d" % (fname, t, rate, capacity, alpha, rf, rp)
    for n in (1, 67, 300) + (((1 << 17) + 3,) if t == 3 and k % 3 == 0 else ()):
        _lib.check(_lib.lib().pmx_ctx_engine_info(ctx._h, _lib.OP_PERMUTE, n, 0, ctypes.byref(info)))
        states = synth.random_elements(f, n * t, seed=k * 7 + n).reshape(n, t, 4)
        states[0] = 0
        if n > 1:
            states[1] = f.from_ints([p - 1] * t)
        if n > 2:
            states[2, 0] = 0
        got, want = ctx.permute_batch(states), cr.permute_batch(states, threads=0)
        if not np.array_equal(got, want):
            bad += 1
            print("PERMUTE differs: %s n=%d engine=%s (%d states)" % (what, n, info.engine.decode(), int((got != want).any(axis=(1, 2)).sum())), flush=True)
    L, ko = rng.randint(0, 2 * t + 1), rng.randint(0, t + 2)
    if L + ko > 0:
        n = 150
        msgs = synth.random_elements(f, max(n * L, 1), seed=k).reshape(n, L, 4) if L else np.zeros((n, 0, 4), dtype=np.uint64)
        try:
            got, want = ctx.hash_batch(msgs, L, ko, n), cr.hash_batch(msgs, L, ko, threads=0)
            if not np.array_equal(got, want):
                bad += 1
                print("HASH differs: %s L=%d k=%d" % (what, L, ko), flush=True)
        except Exception as e:
            print("HASH raised: %s L=%d k=%d: %s" % (what, L, ko, e), flush=True)
            bad += 1
    if rate >= 2:
        leaves = synth.random_elements(f, 128, seed=k + 5)
        nodes, _ = ctx.merkle_2to1(leaves)
        if not np.array_equal(nodes, cr.merkle(leaves, threads=0)):
            bad += 1
            print("MERKLE differs: %s" % what, flush=True)
    # the duplex driver: sponges in random modes and positions advance together, checked sponge by sponge
    if k % 2 == 0 and rate >= 1:
        n, r = 90, rate
        nrng = np.random.default_rng(k)
        batch = S.BatchPoseidonSponge.new(cfg, n)
        batch.state = synth.random_elements(f, n * t, seed=k + 11).reshape(n, t, 4)
        batch.mode_tag = nrng.integers(0, 2, n).astype(np.uint32)
        batch.mode_index = nrng.integers(0, r + 1, n).astype(np.uint32)
        ref = [(batch.state[i].copy(), int(batch.mode_tag[i]), int(batch.mode_index[i])) for i in range(n)]
        for (op, length) in [("absorb", rng.randint(1, 2 * r + 1)), ("squeeze", rng.randint(1, 2 * r + 1)), ("squeeze", 1), ("absorb", 1)]:
            ok = True
            if op == "absorb":
                elems = synth.random_elements(f, n * length, seed=length + k).reshape(n, length, 4)
                batch.absorb(elems)
                ref = [cr.sponge_absorb(st, m, i, elems[j]) for j, (st, m, i) in enumerate(ref)]
            else:
                out = batch.squeeze_native_field_elements(length)
                nxt = []
                for j, (st, m, i) in enumerate(ref):
                    s2, m2, i2, o = cr.sponge_squeeze(st, m, i, length)
                    ok = ok and np.array_equal(out[j], o)
                    nxt.append((s2, m2, i2))
                ref = nxt
            for j, (st, m, i) in enumerate(ref):
                ok = ok and np.array_equal(batch.state[j], st) and (batch.mode_tag[j], batch.mode_index[j]) == (m, i)
            if not ok:
                bad += 1
                print("SPONGE %s(%d) differs: %s" % (op, length, what), flush=True)
    ctx.close() if hasattr(ctx, "close") else None
print("fuzz_configs: %d configurations, %d failing cases, %.0f s" % (N, bad, time.time() - t0))
sys.exit(1 if bad else 0)
