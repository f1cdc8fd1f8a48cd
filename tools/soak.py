"""Soak of the small-launch (quad engine) and mid-size paths and of the wide-state driver passes: many launches of random
sizes, lengths and mode mixes, every result checked against the C restatement (whole batch for small n, a sample above).
usage: python tools/soak.py [seconds] [launch sets]
(No torch in this process: the library then runs on the system's HIP runtime, /opt/rocm, as it does under a Rust or C++ caller -
a process that imports torch first gets torch's bundled runtime instead, and the two have behaved differently.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sponge_amd as S  # noqa: E402
from sponge_amd import synth  # noqa: E402
from oracle import cref  # noqa: E402
from oracle import poseidon_oracle as O  # noqa: E402



def main(budget=60.0, max_iters=None):
    """Runs for `budget` seconds (or `max_iters` launch sets, whichever comes first); returns (launch sets, results checked)."""
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
    cases.append((S.poseidon_config_from_lfsr(f, 4, 5, 8, 56), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 4, 5, 8, 56)), 5, [1, 64, 65, 1000, 5000, 50001]))
    cases.append((S.poseidon_config_from_lfsr(S.BN254_FR, 8, 5, 8, 57), cref.CRef(O.make_config(O.BN254_FR, 254, 8, 5, 8, 57)), 9, [1, 64, 130, 3000, 70001]))
    # the driver passes of the other hybrid widths (round 4), and a field on each side of the matrix-core modulus rule
    cases.append((S.poseidon_config_from_lfsr(f, 3, 5, 8, 56), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 3, 5, 8, 56)), 4, [1, 255, 257, 3000, 40000]))
    cases.append((S.poseidon_config_from_lfsr(f, 5, 3, 8, 57), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 5, 3, 8, 57)), 6, [1, 64, 513, 9000]))
    _pallas, _p25519 = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001, (1 << 255) - 19
    cases.append((S.poseidon_config_from_lfsr(S.Field("pallas_fp", _pallas), 8, 5, 8, 57), cref.CRef(O.make_config(_pallas, 255, 8, 5, 8, 57)), 9, [1, 257, 2000]))
    cases.append((S.poseidon_config_from_lfsr(S.Field("p25519", _p25519), 7, 5, 8, 57), cref.CRef(O.make_config(_p25519, 255, 7, 5, 8, 57)), 8, [1, 257, 2000]))
    cases.append((S.poseidon_config_from_lfsr(f, 11, 5, 8, 57), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 11, 5, 8, 57)), 12, [1, 64, 130]))
    # t = 7, 8: the other widths of the matrix-core engine (table stages of 6 + 2 and 6 + 3 k-steps), alpha 5 and the generic-exponent build
    cases.append((S.poseidon_config_from_lfsr(f, 6, 5, 8, 57), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 6, 5, 8, 57)), 7, [1, 33, 64, 257, 1000, 20000]))
    cases.append((S.poseidon_config_from_lfsr(f, 7, 17, 8, 57), cref.CRef(O.make_config(O.BLS12_381_FR, 255, 7, 17, 8, 57)), 8, [1, 65, 300, 2000]))
    t0, it, checked = time.time(), 0, 0
    while time.time() - t0 < budget and (max_iters is None or it < max_iters):
        cfg, cr, t, sizes = cases[it % len(cases)]
        fld, rate = cfg.field, cfg.rate
        n = int(rng.choice(sizes))
        seed = int(rng.integers(1 << 30))
        b = S.BatchPoseidonSponge.new(cfg, n)
        b.state[:] = synth.random_elements(fld, n * t, seed=seed).reshape(n, t, 4)
        b.mode_tag[:] = rng.integers(0, 2, n, dtype=np.uint32)
        b.mode_index[:] = rng.integers(0, rate + 1, n, dtype=np.uint32)
        st0, tag0, idx0 = b.state.copy(), b.mode_tag.copy(), b.mode_index.copy()
        L, k = int(rng.integers(1, 3 * rate + 3)), int(rng.integers(0, 3 * rate + 3))
        msgs = synth.random_elements(fld, n * L, seed=seed + 1).reshape(n, L, 4)
        b.absorb(msgs)
        got = b.squeeze_native_field_elements(k)
        pick = np.arange(n) if n <= 300 else rng.choice(n, 200, replace=False)
        for i in pick:
            s, m, x = cr.sponge_absorb(st0[i], int(tag0[i]), int(idx0[i]), msgs[i])
            s, m, x, out = cr.sponge_squeeze(s, m, x, k)
            assert np.array_equal(got[i], out) and np.array_equal(b.state[i], s) and (int(b.mode_tag[i]), int(b.mode_index[i])) == (m, x), \
                (it, fld.name, t, n, int(i), "absorb", L, "squeeze", k, "mode", int(tag0[i]), int(idx0[i]), "->", (int(b.mode_tag[i]), int(b.mode_index[i])), "want", (m, x),
                 "out ok", bool(np.array_equal(got[i], out)), "state ok", bool(np.array_equal(b.state[i], s)))
        ps = synth.random_elements(fld, n * t, seed=seed + 2).reshape(n, t, 4)
        gp = cfg.context().permute_batch(ps)
        sub = pick
        assert np.array_equal(gp[sub], cr.permute_batch(np.ascontiguousarray(ps[sub]), threads=0)), (it, n)
        hs = cfg.context().hash_batch(msgs, L, max(k, 1))
        assert np.array_equal(hs[sub], cr.hash_batch(np.ascontiguousarray(msgs[sub]), L, max(k, 1), threads=0)), (it, n)
        checked += 3 * len(pick)
        it += 1
    print("soak ok: %d launches-sets, %d results checked in %.0f s" % (it, checked, time.time() - t0))
    return it, checked


if __name__ == "__main__":      # python tools/soak.py [seconds] [launch sets]
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else None)
