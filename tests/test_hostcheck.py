"""The kernel ALGORITHMS (unsaturated 9x29-bit Montgomery arithmetic, interleaved dot/REDC, round schedule)
compiled for the host and compared with the oracle on CPU.  This exercises the exact templates the HIP
kernels instantiate (sponge_amd/csrc/pmx_field.hpp, pmx_permute.hpp); it is test infrastructure - the
product library has no CPU data path."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

from sponge_amd._lib import PmxConfig
from oracle import cref
from oracle import poseidon_oracle as O

from helpers import golden, ints, oracle_config

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck")


@pytest.fixture(scope="module")
def hc():
    subprocess.check_call(["make", "-C", HERE, "all"], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.environ.get("PMX_HOSTCHECK_LIB") or os.path.join(HERE, "libpmx_hostcheck.so"))
    lib.hc_permute.argtypes = [ctypes.POINTER(PmxConfig), ctypes.c_void_p, ctypes.c_size_t]
    lib.hc_permute_rt.argtypes = [ctypes.POINTER(PmxConfig), ctypes.c_void_p, ctypes.c_size_t]
    lib.hc_permute_coop.argtypes = [ctypes.POINTER(PmxConfig), ctypes.c_void_p, ctypes.c_size_t]
    lib.hc_permute_hybrid_mfma.argtypes = [ctypes.POINTER(PmxConfig), ctypes.c_void_p, ctypes.c_size_t]
    lib.hc_field_op.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.hc_worst_column.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    lib.hc_worst_sqr_column.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    lib.hc_worst_tab_column.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    lib.hc_worst_dense_rt_row.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.hc_tab_op.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def mont_limbs(vals, p):
    return cref.elems_to_limbs(vals, p)


def field_op(hc, p, op, a_vals, b_vals):
    mod = np.array(O.to_limbs(p), dtype=np.uint64)
    a = mont_limbs(a_vals, p)
    b = mont_limbs(b_vals, p)
    out = np.zeros(4, dtype=np.uint64)
    assert hc.hc_field_op(mod.ctypes.data, op, a.ctypes.data, b.ctypes.data, out.ctypes.data) == 0
    raw = O.from_limbs([int(x) for x in out])
    assert raw < p, "result not fully reduced"
    return O.from_mont(raw, p)


@pytest.mark.parametrize("p", [O.BLS12_381_FR, O.BN254_FR])
def test_field_ops_match_bigint(hc, p):
    rng = random.Random(p & 0xFFFF)
    edge = [0, 1, 2, p - 1, p - 2, (1 << 255) % p, (p - 1) // 2, (1 << 29) - 1, 1 << 232]
    pairs = [(a, b) for a in edge for b in edge] + [(rng.randrange(p), rng.randrange(p)) for _ in range(300)]
    for a, b in pairs:
        assert field_op(hc, p, 0, [a], [b]) == a * b % p
        assert field_op(hc, p, 1, [a], [b]) == a * a % p
        assert field_op(hc, p, 3, [a], [b]) == a
    for _ in range(200):
        xs = [rng.choice(edge + [rng.randrange(p)]) for _ in range(3)]
        ys = [rng.choice(edge + [rng.randrange(p)]) for _ in range(3)]
        assert field_op(hc, p, 2, xs, ys) == sum(x * y for x, y in zip(xs, ys)) % p


def test_column_accumulators_cannot_overflow(hc):
    """Worst case over ALL limb patterns: 3-term dot, one side lazily added (limbs < 2^30), the other
    normalised (< 2^29), plus the 9 reduction products and the carry, stays below 2^64."""
    hi = np.zeros(1, dtype=np.uint64)
    lo = np.zeros(1, dtype=np.uint64)
    for terms, amax, bmax, ok in [(3, (1 << 30) - 1, (1 << 29) - 1, True),     # MDS row on x + c
                                  (1, (1 << 30) - 1, (1 << 30) - 1, True),     # product of two lazy operands
                                  (6, (1 << 29) - 1, (1 << 29) - 1, True),     # 6 normalised terms
                                  (7, (1 << 29) - 1, (1 << 29) - 1, False),    # ... is the limit
                                  (4, (1 << 30) - 1, (1 << 30) - 1, False)]:   # and so is this
        hc.hc_worst_column(terms, amax, bmax, hi.ctypes.data, lo.ctypes.data)
        assert (int(hi[0]) == 0) == ok, (terms, amax, bmax, int(hi[0]), int(lo[0]))
    hc.hc_worst_sqr_column((1 << 30) - 1, hi.ctypes.data, lo.ctypes.data)     # squaring a lazily added operand
    assert int(hi[0]) == 0


@pytest.mark.parametrize("p", [O.BLS12_381_FR, O.BN254_FR, (1 << 230) + 0x1D])
def test_shifted_table_products_match_bigint(hc, p):
    """tab_lanes_stream one operation at a time (the history term of a t = 3 window is its single-constant form): s_i + a * c_i through
    the nine shifted residues of c_i plus two Montgomery steps is the same field element as the full product and sum, for one, two and
    eight constants in a row, on edge operands (0, 1, p-1, values with all-ones limbs) and random ones."""
    rng = random.Random(17)
    mod = np.array(O.to_limbs(p), dtype=np.uint64)
    edge = [0, 1, p - 1, p - 2, ((1 << 29) - 1) * sum(1 << (29 * i) for i in range(8)) % p, (1 << 229) - 1]

    def pick():
        return rng.choice(edge) if rng.random() < 0.4 else rng.randrange(p)

    def run(n, a, c, s_):
        out = np.zeros(4 * n, dtype=np.uint64)
        la, lc, ls = (mont_limbs(v, p) for v in (a, c, s_))
        assert hc.hc_tab_op(mod.ctypes.data, n, la.ctypes.data, lc.ctypes.data, ls.ctypes.data, out.ctypes.data) == 0
        return cref.limbs_to_elems(out, p)

    for _ in range(60):
        for n in (1, 2, 8):
            a, c, s_ = pick(), [pick() for _ in range(n)], [pick() for _ in range(n)]
            assert run(n, [a], c, s_) == [(a * ci + si) % p for ci, si in zip(c, s_)], n


def test_run_time_width_rows_cannot_overflow(hc):
    """Rows of the run-time-width engine (lazy operands, explicit 64-bit columns): compression before every fourth term, the tail left to
    the reduction.  Replay of the exact schedule with every limb (operands, p, m) at its maximum: every column stays below 2^64 for
    every row length the engine takes (t <= 16)."""
    hi, lo = np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint64)
    for terms in range(1, 17):
        hc.hc_worst_dense_rt_row(terms, hi.ctypes.data, lo.ctypes.data)
        assert int(hi[0]) == 0, terms


def test_table_column_accumulators_cannot_overflow(hc):
    """Every column of a table product holds nine products per term: six normalised terms (+ 2 reduction products, the
    addend, the carry) fit 64 bits, seven do not (the kernels use one term: the t = 3 history product); a lazily added
    operand (limbs < 2^30) would fit three terms only, which is why the operands must be normalised."""
    hi = np.zeros(1, dtype=np.uint64)
    lo = np.zeros(1, dtype=np.uint64)
    for terms, zmax, ok in [(6, (1 << 29) - 1, True), (7, (1 << 29) - 1, False), (5, (1 << 29) - 1, True),
                            (3, (1 << 30) - 1, True), (4, (1 << 30) - 1, False)]:
        hc.hc_worst_tab_column(terms, zmax, hi.ctypes.data, lo.ctypes.data)
        assert (int(hi[0]) == 0) == ok, (terms, zmax, int(hi[0]))


def run_permute(hc, name, states, coop=False, mfma=False):
    """the host build of one schedule: the reference's dense one at run-time width (what LdsEngine runs; default), the quad engine's
    (coop, t = 3), the window engines' (mfma: HybridEngine<3..9>)"""
    cfg = oracle_config(name)
    p = cfg.p
    ark = mont_limbs([v for row in cfg.ark for v in row], p)
    mds = mont_limbs([v for row in cfg.mds for v in row], p)
    c = PmxConfig()
    c.full_rounds, c.partial_rounds, c.alpha = cfg.full_rounds, cfg.partial_rounds, cfg.alpha
    c.rate, c.capacity = cfg.rate, cfg.capacity
    for i, l in enumerate(O.to_limbs(p)):
        c.modulus[i] = l
    c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
    out = np.ascontiguousarray(states, dtype=np.uint64).copy()
    n = out.size // (cfg.t * 4)
    fn = hc.hc_permute_hybrid_mfma if mfma else hc.hc_permute_coop if coop else hc.hc_permute_rt
    assert fn(ctypes.byref(c), out.ctypes.data, n) == 0
    return out


@pytest.mark.parametrize("name", ["bls_t3_a5_8_31", "bls_t3_a17_8_31", "bls_t4_a5_8_56", "bls_t9_a5_8_57",
                                  "bls_t3_a257_8_13", "bn254_t9_a5_8_57", "bn254_t3_a5_8_57",
                                  "reference_test_a17_8_29"])
def test_permutation_templates_match_golden(hc, name):
    cfg = oracle_config(name)
    vecs = golden("permute_vectors.json")[name]
    states = mont_limbs([x for v in vecs for x in ints(v["in"])], cfg.p).reshape(len(vecs), cfg.t, 4)
    want = [x for v in vecs for x in ints(v["out"])]
    out = run_permute(hc, name, states)                 # the reference's dense schedule at run-time width (LdsEngine)
    assert cref.limbs_to_elems(out, cfg.p) == want, "dense, run-time width"
    if cfg.t == 3:
        out = run_permute(hc, name, states, coop=True)  # one state per quad of lanes (small calls at t = 3)
        assert cref.limbs_to_elems(out, cfg.p) == want, "coop"
    out = run_permute(hc, name, states, mfma=True)      # the window engines: every layer through the int8 tables and the row finish
    assert cref.limbs_to_elems(out, cfg.p) == want, "windows, layers on the int8 tables"


def test_permutation_templates_match_c_oracle_on_random_batch(hc):
    from sponge_amd import synth
    import sponge_amd as S
    for name, f in [("bls_t3_a5_8_31", S.BLS12_381_FR), ("bn254_t3_a5_8_57", S.BN254_FR)]:
        states = synth.random_elements(f, 512 * 3, seed=77).reshape(512, 3, 4)
        want = cref.CRef(oracle_config(name)).permute_batch(states, threads=0)
        assert np.array_equal(run_permute(hc, name, states), want)
        assert np.array_equal(run_permute(hc, name, states, mfma=True), want)
        assert np.array_equal(run_permute(hc, name, states, coop=True), want)
    from sponge_amd import synth as sy
    states = sy.random_elements(S.BN254_FR, 64 * 9, seed=78).reshape(64, 9, 4)
    want = cref.CRef(oracle_config("bn254_t9_a5_8_57")).permute_batch(states, threads=0)
    assert np.array_equal(run_permute(hc, "bn254_t9_a5_8_57", states), want)
    assert np.array_equal(run_permute(hc, "bn254_t9_a5_8_57", states, mfma=True), want)
    # edge states: every element 0, 1 (Montgomery), p - 1
    from oracle import poseidon_oracle as Oo
    p = Oo.BN254_FR
    edge = mont_limbs([v for e in (0, 1, p - 1, p - 2, (1 << 253) + 12345) for v in [e] * 9], p).reshape(5, 9, 4)
    want = cref.CRef(oracle_config("bn254_t9_a5_8_57")).permute_batch(edge, threads=0)
    assert np.array_equal(run_permute(hc, "bn254_t9_a5_8_57", edge, mfma=True), want)


def test_identity_lane_magnitudes_stay_inside_their_bounds(hc):
    """The identity lanes of the quad engine's sparse partial rounds (t = 3) are updated without a magnitude cap (mont_mul_add): they grow
    by at most 1.0204 p per round and must stay below 2^261 (normalised limbs).  The host build reports every lane and every row-0
    output of the partial section: limbs < 2^29, lanes inside the worst-case line pmx_prepare.hpp budgets for (opt_schedule_lane_headroom),
    row 0 small."""
    from sponge_amd import synth
    import sponge_amd as S
    hc.hc_track_reset.argtypes = []
    hc.hc_track_get.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for name in ("bls_t3_a5_8_31", "bn254_t3_a5_8_57"):
        cfg = oracle_config(name)
        assert 2.2 + (1 + 1.3 * cfg.p / (1 << 261)) * (cfg.partial_rounds - 1) + 1.5 < (1 << 261) / cfg.p     # what prepare() checks
    # the quad engine's folded sparse rounds (t = 3, alpha 5 / 17) accumulate their lanes the same way; row 0 there is
    # x^4 (x m00) + (v_1 s_1 + v_2 s_2): below 6 p, limbs normalised
    for name, f in [("bls_t3_a5_8_31", S.BLS12_381_FR), ("bls_t3_a17_8_31", S.BLS12_381_FR), ("bn254_t3_a5_8_57", S.BN254_FR)]:
        cfg = oracle_config(name)
        states = synth.random_elements(f, 96 * 3, seed=6).reshape(96, 3, 4)
        edge = cref.elems_to_limbs([cfg.p - 1] * 3 + [0] * 3 + [1] * 3, cfg.p).reshape(3, 3, 4)
        states = np.concatenate([states, edge])
        hc.hc_track_reset()
        assert np.array_equal(run_permute(hc, name, states, coop=True), cref.CRef(cfg).permute_batch(states, threads=0))
        for tag, limit in [(0, 6.0), (1, 2.2 + 1.0204 * (cfg.partial_rounds - 1))]:
            limb = np.zeros(1, dtype=np.uint32)
            b = np.zeros(1, dtype=np.float64)
            hc.hc_track_get(tag, limb.ctypes.data, b.ctypes.data)
            assert 0 < int(limb[0]) < (1 << 29), (name, tag, int(limb[0]))
            assert 0 < float(b[0]) < limit, (name, tag, float(b[0]))


def test_long_partial_sections_leave_the_optimised_schedule(hc):
    """prepare() keeps the optimised schedule only while the uncapped identity lanes provably stay below 2^261:
    up to 66 partial rounds for BLS12-381 Fr (2^261 / p = 70.66), far more for the 254-bit BN254 Fr."""
    rng = random.Random(3)
    hc.hc_track_reset.argtypes = []
    hc.hc_track_get.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    # alpha = 2 is the S-box whose output is NOT small (z_0 up to 2.7 p): the lanes grow faster and prepare() allows fewer rounds
    # (alpha = 1 is formed as the product x * 1 - pmx_field.hpp: fe_sbox - and is as small as any other)
    for p, bits, rp, alpha, expect_opt in [(O.BLS12_381_FR, 255, 66, 5, True), (O.BLS12_381_FR, 255, 67, 5, False),
                                           (O.BN254_FR, 254, 120, 5, True), (O.BLS12_381_FR, 255, 66, 1, True),
                                           (O.BLS12_381_FR, 255, 67, 1, False), (O.BLS12_381_FR, 255, 64, 2, True),
                                           (O.BLS12_381_FR, 255, 66, 2, False)]:
        cfg = O.make_config(p, bits, 2, alpha, 8, rp)
        states = [[rng.randrange(p) for _ in range(3)] for _ in range(4)] + [[p - 1] * 3]
        want = [x for st in states for x in O.permute(cfg, st)]
        limbs = cref.elems_to_limbs([x for st in states for x in st], p).reshape(len(states), 3, 4)
        ark = cref.elems_to_limbs([v for row in cfg.ark for v in row], p)
        mds = cref.elems_to_limbs([v for row in cfg.mds for v in row], p)
        c = PmxConfig()
        c.full_rounds, c.partial_rounds, c.alpha, c.rate, c.capacity = 8, rp, alpha, 2, 1
        for i, l in enumerate(O.to_limbs(p)):
            c.modulus[i] = l
        c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
        out = limbs.copy()
        hc.hc_track_reset()
        rc = hc.hc_permute_coop(ctypes.byref(c), out.ctypes.data, len(states))     # (the quad engine: the schedule with the uncapped lanes)
        assert (rc == 0) == expect_opt, (bits, rp, alpha, rc)
        out_win = limbs.copy()
        assert (hc.hc_permute_hybrid_mfma(ctypes.byref(c), out_win.ctypes.data, len(states)) == 0) == expect_opt
        if expect_opt:
            assert cref.limbs_to_elems(out, p) == want
            assert cref.limbs_to_elems(out_win, p) == want
            limb = np.zeros(1, dtype=np.uint32)
            b = np.zeros(1, dtype=np.float64)
            hc.hc_track_get(1, limb.ctypes.data, b.ctypes.data)
            assert int(limb[0]) < (1 << 29) and float(b[0]) < (1 << 261) / p - 1.5, (bits, rp, alpha, float(b[0]))
        out = limbs.copy()
        assert hc.hc_permute(ctypes.byref(c), out.ctypes.data, len(states)) == 0     # the dense schedule always works
        assert cref.limbs_to_elems(out, p) == want


PALLAS_FP = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001      # 255 bits, not a reference field
P25519 = (1 << 255) - 19   # top byte 0x7f
SMALL_P = (1 << 230) + 0x1D                                                         # pseudo-modulus near the lower limit


def _odd_modulus_config(p, bits, rate, alpha, rf, rp):
    return O.make_config(p, bits, rate, alpha, rf, rp)


@pytest.mark.parametrize("p", [PALLAS_FP])
def test_other_255_bit_prime(hc, p):
    """Nothing in the arithmetic is specific to the two benchmarked fields: a third 255-bit prime (Pallas base field),
    constants from the same Grain-LFSR procedure, every schedule against the big-integer oracle."""
    from sponge_amd._lib import PmxConfig
    cfg = O.make_config(p, 255, 2, 5, 8, 56)
    rng = random.Random(99)
    states = [[rng.randrange(p) for _ in range(3)] for _ in range(8)] + [[p - 1] * 3, [0, 0, 0]]
    want = [x for st in states for x in O.permute(cfg, st)]
    limbs = cref.elems_to_limbs([x for st in states for x in st], p).reshape(len(states), 3, 4)
    ark = cref.elems_to_limbs([v for row in cfg.ark for v in row], p)
    mds = cref.elems_to_limbs([v for row in cfg.mds for v in row], p)
    c = PmxConfig()
    c.full_rounds, c.partial_rounds, c.alpha, c.rate, c.capacity = 8, 56, 5, 2, 1
    for i, l in enumerate(O.to_limbs(p)):
        c.modulus[i] = l
    c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
    for fn in (hc.hc_permute, hc.hc_permute_rt, hc.hc_permute_hybrid_mfma, hc.hc_permute_coop):
        out = limbs.copy()
        assert fn(ctypes.byref(c), out.ctypes.data, len(states)) == 0
        assert cref.limbs_to_elems(out, p) == want, fn.__name__


def test_matrix_core_tables_for_every_modulus_also_one_whose_residues_exceed_32_balanced_bytes(hc):
    """pmx_mfma.hpp stores residues in 32 balanced signed bytes, which reach 127 (256^32 - 1) / 255 = 0.996 * 2^255: a residue of
    2^255 - 19 (top byte 0x7f) can lie above that and is then stored as Y - p, with 255 p in the row's correction (pmx_prepare.hpp:
    put_mfma_layer_io; until round 6 such moduli had no tables).  Both the matrix-core form and the dense schedule agree with the
    big-integer oracle at t = 9 - 2^255 - 19, and Pallas (top byte 0x40) where no residue needs the second form."""
    for p, ok in ((P25519, True), (PALLAS_FP, True)):
        cfg = O.make_config(p, 255, 8, 5, 8, 57)
        rng = random.Random(4242)
        states = [[rng.randrange(p) for _ in range(9)] for _ in range(3)] + [[p - 1] * 9]
        want = [x for st in states for x in O.permute(cfg, st)]
        limbs = cref.elems_to_limbs([x for st in states for x in st], p).reshape(len(states), 9, 4)
        ark = cref.elems_to_limbs([v for row in cfg.ark for v in row], p)
        mds = cref.elems_to_limbs([v for row in cfg.mds for v in row], p)
        c = PmxConfig()
        c.full_rounds, c.partial_rounds, c.alpha, c.rate, c.capacity = 8, 57, 5, 8, 1
        for i, l in enumerate(O.to_limbs(p)):
            c.modulus[i] = l
        c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
        out = limbs.copy()
        assert hc.hc_permute(ctypes.byref(c), out.ctypes.data, len(states)) == 0
        assert cref.limbs_to_elems(out, p) == want
        out = limbs.copy()
        rc = hc.hc_permute_hybrid_mfma(ctypes.byref(c), out.ctypes.data, len(states))
        assert rc == (0 if ok else -4), (hex(p >> 248), rc)
        if ok:
            assert cref.limbs_to_elems(out, p) == want


@pytest.mark.parametrize("t", [3, 4, 6, 9])
@pytest.mark.parametrize("alpha", [0, 1, 2, 3])
def test_small_exponents_on_every_schedule(hc, t, alpha):
    """The optimised schedules add an S-box output into rows unreduced and the matrix-core form cuts it into 32 bytes: both stand on the
    bound of a Montgomery PRODUCT (below 1.3 p).  alpha = 1 used to hand its lazy input through - wrong results on the hybrid engines of
    t >= 6 - and is formed as the product x * 1 since (pmx_field.hpp: fe_sbox); alpha = 0 is the constant 1.  Every schedule of the host
    build against the oracle (the reference accepts any alpha: src/poseidon/mod.rs:63-74)."""
    for name in ("hc_permute_hybrid_mfma", "hc_permute_rt"):
        getattr(hc, name).argtypes = [ctypes.POINTER(PmxConfig), ctypes.c_void_p, ctypes.c_size_t]
    p, rate = O.BLS12_381_FR, t - 1
    cfg = O.make_config(p, 255, rate, alpha, 8, 57)
    rng = random.Random(10 * t + alpha)
    states = [[rng.randrange(p) for _ in range(t)] for _ in range(3)] + [[p - 1] * t, [0] * t]
    want = [x for st in states for x in O.permute(cfg, st)]
    limbs = cref.elems_to_limbs([x for st in states for x in st], p).reshape(len(states), t, 4)
    ark = cref.elems_to_limbs([v for row in cfg.ark for v in row], p)
    mds = cref.elems_to_limbs([v for row in cfg.mds for v in row], p)
    c = PmxConfig()
    c.full_rounds, c.partial_rounds, c.alpha, c.rate, c.capacity = 8, 57, alpha, rate, 1
    for i, l in enumerate(O.to_limbs(p)):
        c.modulus[i] = l
    c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
    for name in ("hc_permute_hybrid_mfma", "hc_permute_rt") + (("hc_permute",) if t in (3, 4, 9) else ()):
        out = limbs.copy()
        assert getattr(hc, name)(ctypes.byref(c), out.ctypes.data, len(states)) == 0, (name, t, alpha)
        assert cref.limbs_to_elems(out, p) == want, (name, t, alpha)


@pytest.mark.parametrize("bits,p", [(225, 0x1882ad8ca6a206c7cd20288177a3a38f73c4ab461bf14fa62af9de87b),
                                    (240, 0xeb9797ba8095d06d76ced8af9057c675cc48793c0a7424fffabd8db081d7),
                                    (250, 0x2ca0806c3cccc1b9c01d4041d309b8d3b3806aefb11c60086a765dae8b3ccc5)])
@pytest.mark.parametrize("t", [3, 9])
def test_small_moduli_on_every_schedule(hc, bits, p, t):
    """The library takes primes of 225 ... 255 bits (pmx_prepare.hpp).  The bounds of the matrix-core rows are stated for the largest
    (a row is below 2^248 + p) but scale with the modulus - V < (bytes of the inputs) * 255 * p - so the exit's two conditional
    subtractions also do for the smallest.  Every host schedule against the oracle over three primes far below the benchmarked sizes."""
    for name in ("hc_permute_hybrid_mfma", "hc_permute_rt", "hc_permute"):
        getattr(hc, name).argtypes = [ctypes.POINTER(PmxConfig), ctypes.c_void_p, ctypes.c_size_t]
    cfg = O.make_config(p, bits, t - 1, 5, 8, 22)
    rng = random.Random(bits + t)
    states = [[rng.randrange(p) for _ in range(t)] for _ in range(3)] + [[p - 1] * t, [0] * t]
    want = [x for st in states for x in O.permute(cfg, st)]
    limbs = cref.elems_to_limbs([x for st in states for x in st], p).reshape(len(states), t, 4)
    ark = cref.elems_to_limbs([v for row in cfg.ark for v in row], p)
    mds = cref.elems_to_limbs([v for row in cfg.mds for v in row], p)
    c = PmxConfig()
    c.full_rounds, c.partial_rounds, c.alpha, c.rate, c.capacity = 8, 22, 5, t - 1, 1
    for i, l in enumerate(O.to_limbs(p)):
        c.modulus[i] = l
    c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
    for name in ("hc_permute_hybrid_mfma", "hc_permute_rt", "hc_permute"):
        out = limbs.copy()
        assert getattr(hc, name)(ctypes.byref(c), out.ctypes.data, len(states)) == 0, (name, bits, t)
        assert cref.limbs_to_elems(out, p) == want, (name, bits, t)


@pytest.mark.parametrize("t,alpha", [(3, 5), (3, 17), (4, 5), (9, 5)])
def test_a_zero_capacity_lane_skips_its_first_sbox(hc, t, alpha):
    """pmx_permute.hpp (lane0_zero): compress and the first permutation of a hash row know lane 0 is zero, so round 0 puts the config's
    constant S-box(ark'[0][0]) (pmx_prepare.hpp, behind the window tables) there instead of computing it.  The host build takes that
    shortcut whenever lane 0 of a state is zero: such states must still match the oracle."""
    hc.hc_permute_hybrid_mfma.argtypes = [ctypes.POINTER(PmxConfig), ctypes.c_void_p, ctypes.c_size_t]
    p = O.BLS12_381_FR
    rf, rp = 8, {3: 57, 4: 56, 9: 57}[t]
    cfg = O.make_config(p, 255, t - 1, alpha, rf, rp)
    rng = random.Random(100 * t + alpha)
    states = [[0] + [rng.randrange(p) for _ in range(t - 1)] for _ in range(3)] + [[0] * t, [0] + [p - 1] * (t - 1)]
    want = [x for st in states for x in O.permute(cfg, st)]
    limbs = cref.elems_to_limbs([x for st in states for x in st], p).reshape(len(states), t, 4)
    ark = cref.elems_to_limbs([v for row in cfg.ark for v in row], p)
    mds = cref.elems_to_limbs([v for row in cfg.mds for v in row], p)
    c = PmxConfig()
    c.full_rounds, c.partial_rounds, c.alpha, c.rate, c.capacity = rf, rp, alpha, t - 1, 1
    for i, l in enumerate(O.to_limbs(p)):
        c.modulus[i] = l
    c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
    out = limbs.copy()
    assert hc.hc_permute_hybrid_mfma(ctypes.byref(c), out.ctypes.data, len(states)) == 0
    assert cref.limbs_to_elems(out, p) == want


@pytest.mark.parametrize("p,bits,rf,rp,alpha,want", [(O.BLS12_381_FR, 255, 8, 31, 5, [10, 10, 0, 0, 0]), (O.BN254_FR, 254, 8, 57, 5, [19, 0, 19, 0, 0]),
                                                     (O.BLS12_381_FR, 255, 8, 57, 5, [19, 0, 0, 0, 0]), (O.BLS12_381_FR, 255, 8, 31, 17, [10, 0, 0, 0, 0]),
                                                     (O.BN254_FR, 254, 8, 31, 17, [10, 10, 0, 0, 0]), (O.BLS12_381_FR, 255, 8, 13, 257, [4, 0, 0, 0, 0])])
def test_a_windows_free_scale_turns_its_history_constant_into_a_small_integer(hc, p, bits, rf, rp, alpha, want):
    """pmx_prepare.hpp (derive_window_layers): x^_1 of a window may be carried scaled by any lambda; the one history constant of a t = 3
    window becomes h lambda^(alpha^2 - alpha), and where s / h has such a root for s in 1 .. 4 the kernel adds z^_1 s times instead of
    multiplying by a table (pmx_permute.hpp).  The constant depends on the MDS matrix and the exponent only, so a config's windows all
    get the same s or none: BASELINE's C2 config gets 1, BN254 t = 3 (8, 57) gets 2, BN254 with alpha = 17 gets 1; BLS (8, 57), the
    reference's own alpha = 17 and 257 defaults have no such root and keep the table.  The permutation itself against the oracle below."""
    hc.hc_window_small_history.argtypes = [ctypes.POINTER(PmxConfig), ctypes.c_void_p]
    hc.hc_permute_hybrid_mfma.argtypes = [ctypes.POINTER(PmxConfig), ctypes.c_void_p, ctypes.c_size_t]
    cfg = O.make_config(p, bits, 2, alpha, rf, rp)
    ark = cref.elems_to_limbs([v for row in cfg.ark for v in row], p)
    mds = cref.elems_to_limbs([v for row in cfg.mds for v in row], p)
    c = PmxConfig()
    c.full_rounds, c.partial_rounds, c.alpha, c.rate, c.capacity = rf, rp, alpha, 2, 1
    for i, l in enumerate(O.to_limbs(p)):
        c.modulus[i] = l
    c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
    out = np.zeros(5, dtype=np.uint32)
    assert hc.hc_window_small_history(ctypes.byref(c), out.ctypes.data) == 0
    assert out.tolist() == want
    rng = random.Random(rp + alpha)
    states = [[rng.randrange(p) for _ in range(3)] for _ in range(6)] + [[p - 1] * 3, [0] * 3]
    want_states = [x for st in states for x in O.permute(cfg, st)]
    limbs = cref.elems_to_limbs([x for st in states for x in st], p).reshape(len(states), 3, 4)
    assert hc.hc_permute_hybrid_mfma(ctypes.byref(c), limbs.ctypes.data, len(states)) == 0
    assert cref.limbs_to_elems(limbs, p) == want_states


@pytest.mark.parametrize("K", [9, 1, 4, 6])
def test_partial_rounds_as_windows_every_size_and_width(K):
    """pmx_mfma.hpp / pmx_prepare.hpp (derive_window_layers): the matrix-core engines of t = 3..9 run their partial rounds as windows
    of K S-boxes closed by ONE layer each - an exact rewrite derived on the host (basis of the carried lanes chosen so that later
    S-box inputs are sums of an earlier output, a carried coordinate and K - 2 products at most; the first window takes RP mod K).
    Host build of the same templates and tables against the big-integer oracle: the shipped size 9 = the width at most (57 = 3 + 6 x 9
    at t = 9; at t < 9 clamped to t), 1 (every round its own window), 4 (first window of ONE round), 6 (round 4's size); t = 3 .. 9;
    RP = 57, 56, 5, 1; alpha = 5 and the generic-exponent build; both fields of BASELINE."""
    name = "libpmx_hostcheck.so" if K == 9 else "libpmx_hostcheck_k%d.so" % K
    subprocess.check_call(["make", "-C", HERE, name], stdout=subprocess.DEVNULL)
    hc = ctypes.CDLL(os.path.join(HERE, name))
    hc.hc_permute_hybrid_mfma.argtypes = [ctypes.POINTER(PmxConfig), ctypes.c_void_p, ctypes.c_size_t]
    assert hc.hc_mfma_window(9) == K and hc.hc_mfma_window(7) == min(K, 7) and hc.hc_mfma_window(4) == min(K, 4) and hc.hc_mfma_window(3) == min(K, 3) and hc.hc_mfma_window(2) == 0
    cases = [(O.BN254_FR, 254, 8, 5, 8, 57), (O.BLS12_381_FR, 255, 8, 5, 8, 57), (O.BLS12_381_FR, 255, 7, 5, 8, 57),
             (O.BLS12_381_FR, 255, 6, 5, 8, 57), (O.BLS12_381_FR, 255, 8, 17, 8, 56), (O.BN254_FR, 254, 6, 3, 6, 5),
             (O.BLS12_381_FR, 255, 7, 5, 3, 1), (O.BLS12_381_FR, 255, 5, 5, 8, 57), (O.BLS12_381_FR, 255, 4, 5, 8, 56),
             (O.BLS12_381_FR, 255, 3, 5, 8, 56), (O.BN254_FR, 254, 3, 17, 8, 5), (O.BLS12_381_FR, 255, 2, 5, 8, 31),
             (O.BN254_FR, 254, 2, 5, 8, 57), (O.BLS12_381_FR, 255, 2, 5, 3, 2)]
    for p, bits, rate, alpha, rf, rp in cases:
        t = rate + 1
        cfg = O.make_config(p, bits, rate, alpha, rf, rp)
        rng = random.Random(K * 1000 + t * 10 + rp)
        states = [[rng.randrange(p) for _ in range(t)] for _ in range(3)] + [[p - 1] * t, [0] * t]
        want = [x for st in states for x in O.permute(cfg, st)]
        limbs = cref.elems_to_limbs([x for st in states for x in st], p).reshape(len(states), t, 4)
        ark = cref.elems_to_limbs([v for row in cfg.ark for v in row], p)
        mds = cref.elems_to_limbs([v for row in cfg.mds for v in row], p)
        c = PmxConfig()
        c.full_rounds, c.partial_rounds, c.alpha, c.rate, c.capacity = rf, rp, alpha, rate, 1
        for i, l in enumerate(O.to_limbs(p)):
            c.modulus[i] = l
        c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
        out = limbs.copy()
        assert hc.hc_permute_hybrid_mfma(ctypes.byref(c), out.ctypes.data, len(states)) == 0, (K, t, alpha, rf, rp)
        assert cref.limbs_to_elems(out, p) == want, (K, t, alpha, rf, rp)
    # every element that entered a matrix-core layer above - S-box outputs and rows of the layer before, over all these widths, exponents
    # and edge states - was norm and below 2^256: the bound the 32-byte (one k-step per element) form of pmx_mfma.hpp rests on
    seen, too_large = ctypes.c_ulonglong(), ctypes.c_ulonglong()
    hc.hc_layer_inputs(ctypes.byref(seen), ctypes.byref(too_large))
    assert seen.value > 1000 and too_large.value == 0, (seen.value, too_large.value)


@pytest.mark.parametrize("rate", [2, 4])
@pytest.mark.parametrize("rf,rp", [(1, 0), (1, 5), (3, 0), (3, 5), (7, 0), (7, 5), (3, 1), (3, 2)])
def test_odd_full_rounds_follow_the_reference_split(hc, rate, rf, rp):
    """PoseidonConfig::new asserts shapes only (src/poseidon/mod.rs:196-203) and permute runs RF/2 full rounds before the
    partial section and RF - RF/2 after it (:96-116): an odd full_rounds is a legal config.  Every schedule that accepts
    the config (RF = 1 has no full round in front of the partial section, so the optimised tables do not exist and those
    entry points answer PMX_ERR_UNSUPPORTED) against the big-integer oracle."""
    p, t = O.BLS12_381_FR, rate + 1
    cfg = O.make_config(p, 255, rate, 5, rf, rp)
    rng = random.Random(rf * 100 + rp * 10 + rate)
    states = [[rng.randrange(p) for _ in range(t)] for _ in range(6)] + [[p - 1] * t, [0] * t]
    want = [x for st in states for x in O.permute(cfg, st)]
    limbs = cref.elems_to_limbs([x for st in states for x in st], p).reshape(len(states), t, 4)
    ark = cref.elems_to_limbs([v for row in cfg.ark for v in row], p)
    mds = cref.elems_to_limbs([v for row in cfg.mds for v in row], p)
    c = PmxConfig()
    c.full_rounds, c.partial_rounds, c.alpha, c.rate, c.capacity = rf, rp, 5, rate, 1
    for i, l in enumerate(O.to_limbs(p)):
        c.modulus[i] = l
    c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
    has_opt = rf >= 2 and rp >= 1
    fns = [hc.hc_permute, hc.hc_permute_rt, hc.hc_permute_hybrid_mfma]
    if t == 3:
        fns.append(hc.hc_permute_coop)
    for fn in fns:
        out = limbs.copy()
        rc = fn(ctypes.byref(c), out.ctypes.data, len(states))
        if fn in (hc.hc_permute, hc.hc_permute_rt) or has_opt:
            assert rc == 0, (fn.__name__, rc)
            assert cref.limbs_to_elems(out, p) == want, fn.__name__
        else:
            assert rc != 0, fn.__name__


# ---- the absorb / squeeze driver as passes (sponge_amd/csrc/pmx_sponge_plan.hpp) -----------------------------------------------
class _ToySponge(O.PoseidonSponge):
    """The oracle's sponge (mod.rs:121-182, 232-254, 321-341) over a toy permutation that makes the NUMBER and the ORDER
    of permutations visible in the state: every call mixes in a running counter."""

    def _permute(self):
        self.n_perm = getattr(self, "n_perm", 0) + 1
        p = self.cfg.p
        s = self.state
        self.state = [(3 * s[(i + 1) % len(s)] + 7 * s[i] + i + 1) % p for i in range(len(s))]


def _toy_cfg(rate, capacity):
    t = rate + capacity
    return O.PoseidonConfig(O.BLS12_381_FR, 2, 1, 5, [[0] * t for _ in range(3)], [[1] * t for _ in range(t)], rate, capacity)


def _run_passes(hc, cfg, state, tag, index, squeeze, elems_or_len):
    """What pmx_device.hip's launch loop + sponge_first_kernel / permute_listed_kernel do for ONE sponge: pass p moves chunk p-1 in memory, then
    permutes where the plan says; the mode words are read-only until the last pass."""
    hc.hc_sponge_pass.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32,
                                  ctypes.c_size_t, ctypes.c_void_p]
    hc.hc_sponge_passes.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_uint32]
    hc.hc_sponge_passes.restype = ctypes.c_size_t
    length = elems_or_len if squeeze else len(elems_or_len)
    passes = hc.hc_sponge_passes(int(squeeze), length, cfg.rate)
    toy = _ToySponge(cfg, list(state), tag, index)
    out = [None] * length
    n_perm = 0
    plan = np.zeros(5, dtype=np.uint64)
    end = None
    moved = 0
    for p in range(passes):
        hc.hc_sponge_pass(int(squeeze), tag, index, length, cfg.rate, cfg.capacity, p, plan.ctypes.data)
        permute, pos, first, count, end_index = (int(v) for v in plan)
        assert (count == 0 or first == moved) and pos + count <= cfg.t, (p, plan)
        for j in range(count):
            if squeeze:
                out[first + j] = toy.state[pos + j]
            else:
                toy.state[pos + j] = (toy.state[pos + j] + elems_or_len[first + j]) % cfg.p
        moved += count
        if p == passes - 1:
            assert not permute, "the last pass only moves data and rewrites the mode words"
            end = end_index
        elif permute:
            toy._permute()
            n_perm += 1
    assert moved == length
    return toy.state, end, out, n_perm


@pytest.mark.parametrize("rate,capacity", [(1, 1), (2, 1), (3, 2), (8, 1)])
def test_sponge_pass_plan_reproduces_the_reference_driver(hc, rate, capacity):
    """Every (mode, index, length) of absorb and squeeze: the pass plan must perform the same permutations in the same
    order between the same data movements as the reference's absorb / absorb_internal / squeeze_native_field_elements /
    squeeze_internal - the `:175` test, the lazy permutation of a rate filled exactly, squeeze(0) of an absorbing sponge,
    an index equal to the rate, all included."""
    cfg = _toy_cfg(rate, capacity)
    rng = random.Random(rate * 16 + capacity)
    for tag in (O.ABSORBING, O.SQUEEZING):
        for index in range(rate + 1):
            for length in range(0, 3 * rate + 3):
                state = [rng.randrange(cfg.p) for _ in range(cfg.t)]
                elems = [rng.randrange(cfg.p) for _ in range(length)]
                ref = _ToySponge(cfg, list(state), tag, index)
                ref.absorb(elems)
                if length == 0:
                    assert hc.hc_sponge_passes(0, 0, rate) == 0          # an empty absorb is no call at all (mod.rs:234-236)
                else:
                    got_state, end, _, n_perm = _run_passes(hc, cfg, state, tag, index, False, elems)
                    assert got_state == ref.state and n_perm == getattr(ref, "n_perm", 0), ("absorb", tag, index, length)
                    assert (ref.mode, ref.index) == (O.ABSORBING, end), ("absorb", tag, index, length)
                ref = _ToySponge(cfg, list(state), tag, index)
                want = ref.squeeze_native_field_elements(length)
                got_state, end, out, n_perm = _run_passes(hc, cfg, state, tag, index, True, length)
                assert out == want and got_state == ref.state and n_perm == getattr(ref, "n_perm", 0), ("squeeze", tag, index, length)
                assert (ref.mode, ref.index) == (O.SQUEEZING, end), ("squeeze", tag, index, length)


@pytest.mark.parametrize("p", [O.BLS12_381_FR, O.BN254_FR])
def test_absorb_addition_on_abi_residues(hc, p):
    """state[capacity + i] += element of the pass kernel: both operands are fully reduced Montgomery residues; their sum,
    reduced exactly, is the residue of the sum (mod.rs:128,143) - no multiplication involved."""
    rng = random.Random(p & 0xFFF)
    mod = np.array(O.to_limbs(p), dtype=np.uint64)
    cases = [(0, 0), (p - 1, p - 1), (p - 1, 1), (1, p - 1), (p - 1, 0), ((p - 1) // 2, (p + 1) // 2)]
    cases += [(rng.randrange(p), rng.randrange(p)) for _ in range(200)]
    for a, b in cases:
        aa = np.array(O.to_limbs(a), dtype=np.uint64)      # raw residues: the ABI form IS the operand here
        bb = np.array(O.to_limbs(b), dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        assert hc.hc_field_op(mod.ctypes.data, 4, aa.ctypes.data, bb.ctypes.data, out.ctypes.data) == 0
        assert O.from_limbs([int(x) for x in out]) == (a + b) % p
    # unreduced device-resident data (the host entry points reject it; `_dev` callers own their buffers): the carry out of the 256-bit
    # sum takes part in the select, so a sum of 2^256 or more is still reduced once - congruent to a + b mod p like the per-lane kernels'
    # arithmetic, never the bare low 256 bits of the sum
    for a, b in [(2**256 - 1, 2**256 - 1), (2**256 - 1, 1), (2**255, 2**255), (p + 5, 2**256 - p)]:
        aa = np.array(O.to_limbs(a), dtype=np.uint64)
        bb = np.array(O.to_limbs(b), dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        assert hc.hc_field_op(mod.ctypes.data, 4, aa.ctypes.data, bb.ctypes.data, out.ctypes.data) == 0
        got = O.from_limbs([int(x) for x in out])
        assert got == (a + b - p) % 2**256, (a, b, got)            # one subtraction of p, whatever the carry
        assert got % p == (a + b) % p or a + b - p >= 2**256       # ... which is congruent to a + b whenever the difference fits 256 bits
