"""Soak of the small-launch (quad engine) and mid-size paths: many launches of random sizes and mode mixes, every result
checked against the C restatement (whole batch for small n, a sample above).  usage: python tools/soak.py [seconds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sponge_amd as S  # noqa: E402
from sponge_amd import synth  # noqa: E402
from oracle import cref  # noqa: E402
from oracle import poseidon_oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(12345)
f = S.BLS12_381_FR
cases = []      # (product config, checker, width, sizes)
small = [1, 2, 3, 63, 64, 65, 255, 1000, 4097, 32768, 32769, 40000]
for alpha in (5, 17, 257):
    rf, rp = (8, 13) if alpha == 257 else (8, 31)
    cases.append((S.poseidon_config_from_lfsr(f, 2, alpha, rf, rp), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 2, alpha, rf, rp)), 3,
                  small + ([131072 + 77] if alpha == 5 else [])))          # + the table-form driver kernels
# odd full_rounds, wide states (hybrid engines, t = 5 on streamed tables, t = 9 on BN254), run-time width
cases.append((S.poseidon_config_from_lfsr(f, 2, 5, 7, 31), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 2, 5, 7, 31)), 3, small))
cases.append((S.poseidon_config_from_lfsr(f, 4, 5, 8, 56), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 4, 5, 8, 56)), 5, [1, 64, 65, 1000, 5000]))
cases.append((S.poseidon_config_from_lfsr(S.BN254_FR, 8, 5, 8, 57), cref.CRef(O.make_config(O.BN254_FR, 254, 8, 5, 8, 57)), 9, [1, 64, 130, 3000]))
cases.append((S.poseidon_config_from_lfsr(f, 11, 5, 8, 57), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 11, 5, 8, 57)), 12, [1, 64, 130]))
# t = 7, 8: the other widths of the matrix-core engine (table stages of 6 + 2 and 6 + 3 k-steps), alpha 5 and the generic-exponent build
cases.append((S.poseidon_config_from_lfsr(f, 6, 5, 8, 57), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 6, 5, 8, 57)), 7, [1, 33, 64, 257, 1000]))
cases.append((S.poseidon_config_from_lfsr(f, 7, 17, 8, 57), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 7, 17, 8, 57)), 8, [1, 65, 300, 2000]))
t0, it, checked = time.time(), 0, 0
while time.time() - t0 < budget:
    cfg, cr, t, sizes = cases[it % len(cases)]
    fld, rate = cfg.field, cfg.rate
    n = int(rng.choice(sizes))
    seed = int(rng.integers(1 << 30))
    b = S.BatchPoseidonSponge.new(cfg, n)
    b.state[:] = synth.random_elements(fld, n * t, seed=seed).reshape(n, t, 4)
    b.mode_tag[:] = rng.integers(0, 2, n, dtype=np.uint32)
    b.mode_index[:] = rng.integers(0, rate + 1, n, dtype=np.uint32)
    st0, tag0, idx0 = b.state.copy(), b.mode_tag.copy(), b.mode_index.copy()
    L, k = int(rng.integers(1, 2 * rate + 3)), int(rng.integers(0, 2 * rate + 3))
    msgs = synth.random_elements(fld, n * L, seed=seed + 1).reshape(n, L, 4)
    b.absorb(msgs)
    got = b.squeeze_native_field_elements(k)
    pick = np.arange(n) if n <= 300 else rng.choice(n, 200, replace=False)
    for i in pick:
        s, m, x = cr.sponge_absorb(st0[i], int(tag0[i]), int(idx0[i]), msgs[i])
        s, m, x, out = cr.sponge_squeeze(s, m, x, k)
        assert np.array_equal(got[i], out) and np.array_equal(b.state[i], s) and (int(b.mode_tag[i]), int(b.mode_index[i])) == (m, x), (it, n, i)
    ps = synth.random_elements(fld, n * t, seed=seed + 2).reshape(n, t, 4)
    gp = cfg.context().permute_batch(ps)
    sub = pick
    assert np.array_equal(gp[sub], cr.permute_batch(np.ascontiguousarray(ps[sub]), threads=0)), (it, n)
    hs = cfg.context().hash_batch(msgs, L, max(k, 1))
    assert np.array_equal(hs[sub], cr.hash_batch(np.ascontiguousarray(msgs[sub]), L, max(k, 1), threads=0)), (it, n)
    checked += 3 * len(pick)
    it += 1
print("soak ok: %d launches-sets, %d results checked in %.0f s" % (it, checked, time.time() - t0))
