"""Parity of the HIP path (through the C ABI) with the oracle: golden fixtures, the reference's own KAT,
seeded random batches against the C restatement, ragged sizes, mixed modes.  Bit-exact: every comparison
is limb-for-limb equality of Montgomery residues."""
import numpy as np
import pytest

import sponge_amd as S
from sponge_amd import _lib, synth
from oracle import kats as K

from gpu_helpers import c_oracle, ints, product_config
from helpers import golden, oracle_config

pytestmark = pytest.mark.gpu

ALL_CONFIGS = ["bls_t3_a5_8_31", "bls_t3_a17_8_31", "bls_t4_a5_8_56", "bls_t9_a5_8_57", "bls_t3_a257_8_13",
               "bn254_t9_a5_8_57", "bn254_t3_a5_8_57", "reference_test_a17_8_29"]


def test_poseidon_sponge_consistency():
    """The reference's own KAT, src/poseidon/mod.rs:376-399, written the way the reference writes it."""
    Fr = S.BLS12_381_FR
    sponge_param = S.get_default_poseidon_parameters(Fr, 2, False)
    sponge = S.PoseidonSponge.new(sponge_param)
    sponge.absorb(Fr.from_ints([0, 1, 2]))
    res = sponge.squeeze_native_field_elements(3)
    assert Fr.to_ints(res) == K.SPONGE_CONSISTENCY_OUTPUT
    assert sponge.mode == S.DuplexSpongeMode.Squeezing(1)


@pytest.mark.parametrize("name", ALL_CONFIGS)
def test_permute_golden_vectors(name):
    cfg = product_config(name)
    f = cfg.field
    vecs = golden("permute_vectors.json")[name]
    states = f.from_ints([x for v in vecs for x in ints(v["in"])]).reshape(len(vecs), cfg.t, 4)
    out = cfg.context().permute_batch(states)
    assert f.to_ints(out) == [x for v in vecs for x in ints(v["out"])]


@pytest.mark.parametrize("name", ["bls_t3_a5_8_31", "bls_t3_a17_8_31", "bn254_t9_a5_8_57", "bls_t4_a5_8_56"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000])
def test_permute_ragged_batches_vs_c_oracle(name, n):
    cfg = product_config(name)
    states = synth.random_elements(cfg.field, n * cfg.t, seed=1234 + n).reshape(n, cfg.t, 4)
    got = cfg.context().permute_batch(states)
    want = c_oracle(name).permute_batch(states, threads=0)
    assert np.array_equal(got, want)


def test_permute_empty_batch_is_a_no_op():
    cfg = product_config("bls_t3_a5_8_31")
    out = cfg.context().permute_batch(np.zeros((0, 3, 4), dtype=np.uint64))
    assert out.shape == (0, 3, 4)


@pytest.mark.parametrize("name", ["bls_t3_a5_8_31", "bls_t3_a17_8_31", "reference_test_a17_8_29", "bn254_t9_a5_8_57"])
def test_sponge_traces(name):
    """Every branch of absorb / squeeze incl. the mod.rs:175 quirk, via the PoseidonSponge mirror."""
    cfg = product_config(name)
    f = cfg.field
    for tname, steps in golden("sponge_traces.json")[name].items():
        sponge = S.PoseidonSponge.new(cfg)
        for st in steps:
            if st["op"] == "absorb":
                sponge.absorb(f.from_ints(ints(st["in"])))
            else:
                out = sponge.squeeze_native_field_elements(st["n"])
                assert f.to_ints(out) == ints(st["out"]), (name, tname)
            assert f.to_ints(sponge.state) == ints(st["state"]), (name, tname)
            assert [sponge.mode.tag, sponge.mode.index] == st["mode"], (name, tname)


def test_hash_and_merkle_golden():
    g = golden("hash_merkle_vectors.json")
    for name, d in g.items():
        cfg = product_config(name)
        f = cfg.field
        for row in d["hash"]:
            msg = f.from_ints(ints(row["in"])).reshape(1, row["L"], 4)
            out = cfg.context().hash_batch(msg, row["L"], row["k"], n=1)
            assert f.to_ints(out) == ints(row["out"]), (name, row["L"], row["k"])
    cfg = product_config("bls_t3_a5_8_31")
    levels = g["bls_t3_a5_8_31"]["merkle16"]
    nodes, root = cfg.context().merkle_2to1(cfg.field.from_ints(ints(levels[0])))
    assert cfg.field.to_ints(nodes) == [x for lvl in levels for x in ints(lvl)]
    assert cfg.field.to_ints(root.reshape(1, 4)) == ints(levels[-1])


@pytest.mark.parametrize("name,L,k", [("bls_t3_a5_8_31", 2, 1), ("bls_t3_a5_8_31", 7, 5), ("bls_t3_a17_8_31", 3, 3),
                                      ("bn254_t9_a5_8_57", 8, 1), ("bn254_t9_a5_8_57", 19, 10), ("bls_t4_a5_8_56", 4, 4),
                                      ("bn254_t9_a5_8_57", 2, 1), ("bls_t4_a5_8_56", 2, 1)])   # (2, 1) takes the 2-to-1 launcher
def test_hash_batch_vs_c_oracle(name, L, k):
    cfg = product_config(name)
    n = 333
    msgs = synth.random_elements(cfg.field, n * L, seed=99).reshape(n, L, 4)
    got = cfg.context().hash_batch(msgs, L, k)
    want = c_oracle(name).hash_batch(msgs, L, k, threads=0)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name", ["bls_t3_a5_8_31", "bn254_t9_a5_8_57"])
def test_batch_sponge_with_mixed_modes_vs_c_oracle(name):
    """n sponges in different modes and positions advance together; each is checked against the C
    restatement run sponge by sponge."""
    cfg = product_config(name)
    cr = c_oracle(name)
    n, r = 200, cfg.rate
    rng = np.random.default_rng(5)
    batch = S.BatchPoseidonSponge.new(cfg, n)
    batch.state = synth.random_elements(cfg.field, n * cfg.t, seed=7).reshape(n, cfg.t, 4)
    batch.mode_tag = rng.integers(0, 2, n).astype(np.uint32)
    batch.mode_index = rng.integers(0, r + 1, n).astype(np.uint32)
    ref = [(batch.state[i].copy(), int(batch.mode_tag[i]), int(batch.mode_index[i])) for i in range(n)]
    for (op, length) in [("absorb", r + 1), ("squeeze", r), ("squeeze", 1), ("absorb", 1), ("squeeze", 2 * r + 1)]:
        if op == "absorb":
            elems = synth.random_elements(cfg.field, n * length, seed=length).reshape(n, length, 4)
            batch.absorb(elems)
            ref = [cr.sponge_absorb(s, m, i, elems[j]) for j, (s, m, i) in enumerate(ref)]
        else:
            out = batch.squeeze_native_field_elements(length)
            nxt = []
            for j, (s, m, i) in enumerate(ref):
                s2, m2, i2, o = cr.sponge_squeeze(s, m, i, length)
                assert np.array_equal(out[j], o), (op, length, j)
                nxt.append((s2, m2, i2))
            ref = nxt
        for j, (s, m, i) in enumerate(ref):
            assert np.array_equal(batch.state[j], s) and (batch.mode_tag[j], batch.mode_index[j]) == (m, i), (op, j)


def test_squeeze_bytes_bits_and_state_roundtrip():
    """src/poseidon/tests.rs:71-85 (native cast) and the SpongeExt round trip, plus byte/bit truncation."""
    from oracle import poseidon_oracle as O
    cfg = product_config("reference_test_a17_8_29")
    f = cfg.field
    sponge1 = S.PoseidonSponge.new(cfg)
    sponge1.absorb(f.from_ints([123456789]))
    sponge2 = sponge1.clone()
    assert np.array_equal(sponge1.squeeze_native_field_elements(5), sponge2.squeeze_field_elements(5))
    sponge3 = S.PoseidonSponge.from_state(sponge1.into_state(), cfg)
    osp = O.PoseidonSponge(oracle_config("reference_test_a17_8_29"))
    osp.absorb([123456789])
    osp.squeeze_native_field_elements(5)
    a, b = osp.clone(), osp.clone()
    assert sponge3.squeeze_bytes(100) == a.squeeze_bytes(100, 255)
    assert sponge1.squeeze_bits(600) == [bool(x) for x in b.squeeze_bits(600, 255)]


def test_single_field_element_changes_the_output():
    """src/poseidon/tests.rs:26-33 single_field_element / :35-43 list_with_constant_size_element."""
    cfg = product_config("reference_test_a17_8_29")
    f = cfg.field
    lst1 = synth.random_elements(f, 1024 * 8, seed=42)
    lst2 = lst1.copy()
    lst2[3] = f.from_ints([f.to_ints(lst1[3:4])[0] + 1])[0]
    out = cfg.context().hash_batch(np.stack([lst1, lst2]), 1024 * 8, 3)
    assert not np.array_equal(out[0], out[1])
    want = c_oracle("reference_test_a17_8_29").hash_batch(np.stack([lst1, lst2]), 1024 * 8, 3)
    assert np.array_equal(out, want)


def test_bad_arguments_fail_loudly():
    cfg = product_config("bls_t3_a5_8_31")
    ctx = cfg.context()
    with pytest.raises(S.PmxError):      # not a power of two
        ctx.merkle_2to1(np.zeros((3, 4), dtype=np.uint64))
    b = S.BatchPoseidonSponge.new(cfg, 2)
    b.mode_index[1] = 7                  # > rate
    with pytest.raises(S.PmxError):
        b.squeeze_native_field_elements(1)
    b.mode_index[1] = 0
    b.mode_tag[0] = 9
    with pytest.raises(S.PmxError):
        b.absorb(np.zeros((2, 1, 4), dtype=np.uint64))
    # sizes that cannot be real are refused before anything is allocated or launched
    from sponge_amd import _lib
    one = np.zeros((1, 4), dtype=np.uint64)
    with pytest.raises(S.PmxError, match="overflows"):
        _lib.check(_lib.lib().pmx_hash_batch(ctx._h, one.ctypes.data, 1 << 60, one.ctypes.data, 1, 1 << 20))
    with pytest.raises(S.PmxError, match="too large"):
        _lib.check(_lib.lib().pmx_permute_batch(ctx._h, one.ctypes.data, 1 << 40))
    # the context is still usable afterwards
    st = synth.random_elements(cfg.field, 3, seed=1).reshape(1, 3, 4)
    assert np.array_equal(ctx.permute_batch(st), ctx.permute_batch(st.copy()))


@pytest.mark.parametrize("rate", [3, 4, 5, 6, 7, 8])
def test_default_table_widths_vs_c_oracle(rate):
    """Every width of the reference's default table (src/test.rs:14-22, constraints-optimised, alpha = 5): the
    register/LDS hybrid engines on the optimised schedule against the dense C restatement."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.BLS12_381_FR
    cfg = S.get_default_poseidon_parameters(f, rate, False)
    ocfg = O.default_bls12_381_config(rate, False)
    assert (cfg.alpha, cfg.full_rounds, cfg.partial_rounds) == (ocfg.alpha, ocfg.full_rounds, ocfg.partial_rounds)
    t = rate + 1
    for n in (1, 65, 200):
        states = synth.random_elements(f, n * t, seed=1000 * rate + n).reshape(n, t, 4)
        got = cfg.context().permute_batch(states)
        want = cref.CRef(ocfg).permute_batch(states, threads=0)
        assert np.array_equal(got, want), (rate, n)
    msgs = synth.random_elements(f, 100 * (rate + 2), seed=rate).reshape(100, rate + 2, 4)
    assert np.array_equal(cfg.context().hash_batch(msgs, rate + 2, 2), cref.CRef(ocfg).hash_batch(msgs, rate + 2, 2, threads=0))


@pytest.mark.parametrize("name", ["bls_t3_a5_8_31", "bls_t3_a17_8_31", "bls_t3_a257_8_13", "bn254_t3_a5_8_57",
                                  "reference_test_a17_8_29", "bls_t4_a5_8_56", "bn254_t9_a5_8_57"])
def test_merkle_trees_vs_c_oracle(name):
    """2-to-1 trees through both compression kernels (one lane per state for wide levels, the cooperative
    quad-per-state kernel for levels of <= 32768 nodes) for every S-box variant."""
    cfg = product_config(name)
    cr = c_oracle(name)
    for m in (2, 64, 4096):
        leaves = synth.random_elements(cfg.field, m, seed=m)
        nodes, root = cfg.context().merkle_2to1(leaves)
        want = cr.merkle(leaves, threads=0)
        assert np.array_equal(nodes, want), (name, m)
        assert np.array_equal(root, want[-1])


@pytest.mark.parametrize("rate,alpha,rf,rp", [(3, 257, 8, 13), (8, 257, 8, 13), (1, 5, 8, 56), (11, 5, 8, 57), (4, 17, 8, 30),
                                               (2, 3, 8, 10), (4, 2, 4, 5), (2, 1, 2, 3), (1, 0, 2, 2), (2, 5, 8, 0), (2, 7, 0, 9),
                                               (5, 0xFFFFFFFFFFFFFFFF, 2, 2), (15, 5, 4, 6),
                                               (2, 5, 8, 66), (2, 5, 8, 90), (4, 5, 8, 67), (8, 5, 8, 64)])
def test_run_time_width_engine_vs_c_oracle(rate, alpha, rf, rp):
    """Widths, exponents and round splits off the beaten path: weights-optimised table (alpha = 257), t = 2 and t = 12
    (run-time-width engine), alpha = 17 at t = 5, degenerate exponents 0..3 and 2^64-1, no partial rounds, no full rounds
    (dense schedule), and partial sections at / beyond what the optimised schedule's uncapped identity lanes allow
    for BLS12-381 Fr (66 rounds: still optimised; 67 and 90: dense schedule) - all against the C port."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.BLS12_381_FR
    cfg = S.poseidon_config_from_lfsr(f, rate, alpha, rf, rp)
    ocfg = O.make_config(O.BLS12_381_FR, 255, rate, alpha, rf, rp)
    t = rate + 1
    cr = cref.CRef(ocfg)
    for n in (1, 64, 130):
        states = synth.random_elements(f, n * t, seed=77 * rate + n).reshape(n, t, 4)
        assert np.array_equal(cfg.context().permute_batch(states), cr.permute_batch(states, threads=0)), (rate, n)
    L = 2 * rate + 1
    msgs = synth.random_elements(f, 70 * L, seed=rate).reshape(70, L, 4)
    assert np.array_equal(cfg.context().hash_batch(msgs, L, rate + 1), cr.hash_batch(msgs, L, rate + 1, threads=0))
    if rate >= 2:
        leaves = synth.random_elements(f, 128, seed=5)
        nodes, _ = cfg.context().merkle_2to1(leaves)
        assert np.array_equal(nodes, cr.merkle(leaves, threads=0))


@pytest.mark.parametrize("rate", [2, 4, 8, 11])
@pytest.mark.parametrize("rf", [1, 3, 7])
@pytest.mark.parametrize("rp", [0, 5])
def test_odd_full_rounds_vs_c_oracle(rate, rf, rp):
    """The reference accepts an odd full_rounds (PoseidonConfig::new asserts shapes only, src/poseidon/mod.rs:196-203) and
    runs RF/2 full rounds before the partial section, RF - RF/2 after it (:96-116).  Every engine (registers / quad for
    t = 3, hybrid for t = 5 and 9, run-time width for t = 12; dense schedule where RF/2 = 0) through permutation, hash
    driver and tree, sizes on both sides of the small-launch switches, against the C port."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.BLS12_381_FR
    cfg = S.poseidon_config_from_lfsr(f, rate, 5, rf, rp)
    cr = cref.CRef(O.make_config(O.BLS12_381_FR, 255, rate, 5, rf, rp))
    t = rate + 1
    sizes = (1, 130, (1 << 15) + 70, (1 << 17) + 3) if rate == 2 else (1, 130)
    for n in sizes:
        states = synth.random_elements(f, n * t, seed=31 * rf + rp + n).reshape(n, t, 4)
        assert np.array_equal(cfg.context().permute_batch(states), cr.permute_batch(states, threads=0)), (rate, rf, rp, n)
    L = rate + 2
    msgs = synth.random_elements(f, 70 * L, seed=rf).reshape(70, L, 4)
    assert np.array_equal(cfg.context().hash_batch(msgs, L, 2), cr.hash_batch(msgs, L, 2, threads=0))
    leaves = synth.random_elements(f, 256, seed=rp)
    nodes, _ = cfg.context().merkle_2to1(leaves)
    assert np.array_equal(nodes, cr.merkle(leaves, threads=0))


@pytest.mark.parametrize("field_name,modulus", [("pallas_fp", 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001),
                                                ("p25519", (1 << 255) - 19)])
def test_t9_layers_on_the_matrix_cores_for_moduli_on_both_sides_of_the_32_byte_range(field_name, modulus):
    """pmx_mfma.hpp stores a layer's constants in 32 balanced signed bytes, which reach 0.996 * 2^255: every residue of Pallas (top byte
    0x40), but not every residue of 2^255 - 19 (0x7f), whose larger ones are stored as Y - p with 255 p in the row's correction
    (pmx_prepare.hpp: put_mfma_layer_io; until round 6 such moduli ran on VALU rows).  Both - neither is a benchmarked field - on the
    window engine, through permutation (sizes that leave a workgroup, a wave and a lane pair l / l + 32 partly empty), hash driver and
    compression, against the C port."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.Field(field_name, modulus)
    cfg = S.poseidon_config_from_lfsr(f, 8, 5, 8, 57)
    cr = cref.CRef(O.make_config(modulus, 255, 8, 5, 8, 57))
    import ctypes
    info = _lib.PmxEngineInfo()
    _lib.check(_lib.lib().pmx_ctx_engine_info(cfg.context()._h, _lib.OP_PERMUTE, 1000, 0, ctypes.byref(info)))
    assert info.engine == b"HybridEngine<9,5,mfma,windows of 9>" and info.mfma_dense == 1, info.engine
    for n in (1, 33, 70, 255, 257, 1000):
        states = synth.random_elements(f, n * 9, seed=900 + n).reshape(n, 9, 4)
        assert np.array_equal(cfg.context().permute_batch(states), cr.permute_batch(states, threads=0)), (field_name, n)
    edge = f.from_ints([v for e in (0, 1, modulus - 1, modulus - 2) for v in [e] * 9]).reshape(4, 9, 4)
    assert np.array_equal(cfg.context().permute_batch(edge), cr.permute_batch(edge, threads=0))
    msgs = synth.random_elements(f, 70 * 19, seed=5).reshape(70, 19, 4)
    assert np.array_equal(cfg.context().hash_batch(msgs, 19, 10), cr.hash_batch(msgs, 19, 10, threads=0))
    leaves = synth.random_elements(f, 512, seed=6)
    nodes, _ = cfg.context().merkle_2to1(leaves)
    assert np.array_equal(nodes, cr.merkle(leaves, threads=0))
    # the absorb / squeeze drivers (passes on the same engines, tests/test_gpu_sponge_passes.py): new; absorb(11); squeeze(9) = a hash row
    msgs = synth.random_elements(f, 130 * 11, seed=7).reshape(130, 11, 4)
    b = S.BatchPoseidonSponge.new(cfg, 130)
    b.absorb(msgs)
    assert np.array_equal(b.squeeze_native_field_elements(9), cr.hash_batch(msgs, 11, 9, threads=0))


@pytest.mark.parametrize("rate", [6, 7, 8])
@pytest.mark.parametrize("alpha", [5, 17, 3])
def test_widths_7_to_9_and_every_exponent_on_the_matrix_cores(rate, alpha):
    """t = 7, 8, 9 x the dedicated chain (5), the other dedicated one (17: the generic-exponent build of the hybrid engines) and
    a plain square-and-multiply exponent: permutation, hash driver (last permutation computes only the wanted rows) and
    compression against the C port."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.BLS12_381_FR
    cfg = S.poseidon_config_from_lfsr(f, rate, alpha, 8, 57)
    cr = cref.CRef(O.make_config(O.BLS12_381_FR, 255, rate, alpha, 8, 57))
    t = rate + 1
    for n in (1, 95, 300):
        states = synth.random_elements(f, n * t, seed=70 * rate + alpha + n).reshape(n, t, 4)
        assert np.array_equal(cfg.context().permute_batch(states), cr.permute_batch(states, threads=0)), (rate, alpha, n)
    L = 2 * rate + 1
    msgs = synth.random_elements(f, 70 * L, seed=alpha).reshape(70, L, 4)
    assert np.array_equal(cfg.context().hash_batch(msgs, L, 3), cr.hash_batch(msgs, L, 3, threads=0))
    leaves = synth.random_elements(f, 128, seed=rate)
    nodes, _ = cfg.context().merkle_2to1(leaves)
    assert np.array_equal(nodes, cr.merkle(leaves, threads=0))


def test_merkle_tree_paths():
    """Tree container + batch path verification (2-to-1 compression mode) against the oracle's tree."""
    from oracle import poseidon_oracle as O
    cfg = product_config("bls_t3_a5_8_31")
    f = cfg.field
    m = 64
    leaves = synth.random_elements(f, m, seed=11)
    tree = S.MerkleTree(cfg, leaves)
    levels = O.merkle_levels(oracle_config("bls_t3_a5_8_31"), f.to_ints(leaves))
    assert f.to_ints(tree.root.reshape(1, 4)) == levels[-1]
    idx = [0, 1, 37, 63]
    paths = np.stack([tree.path(i) for i in idx])
    for i, p in zip(idx, paths):                      # siblings bottom-up
        want = [levels[l][(i >> l) ^ 1] for l in range(6)]
        assert f.to_ints(p) == want
    ok = S.verify_paths(cfg, leaves[idx], idx, paths, tree.root)
    assert ok.all()
    bad = paths.copy()
    bad[2, 3, 0] ^= np.uint64(1)                      # corrupt one sibling of the third path
    assert list(S.verify_paths(cfg, leaves[idx], idx, bad, tree.root)) == [True, True, False, True]
    assert not S.verify_paths(cfg, leaves[[1, 0]], [0, 1], paths[:2], tree.root).any()   # wrong leaves


def test_many_paths_verify_on_the_device():
    """pmx_merkle_verify_paths keeps the k running nodes on the device for all `depth` levels (one upload, one download):
    4096 paths of a 2^16-leaf tree built by the oracle-checked tree kernel, with corrupted siblings, wrong leaves and
    indices that name no leaf mixed in; the expected verdicts come from the oracle's own walk up each path."""
    import ctypes
    from sponge_amd import _lib
    cfg = product_config("bls_t3_a5_8_31")
    cr = c_oracle("bls_t3_a5_8_31")
    f = cfg.field
    depth, k = 16, 4096
    m = 1 << depth
    leaves = synth.random_elements(f, m, seed=0x5EED0060)
    tree = S.MerkleTree(cfg, leaves)
    assert np.array_equal(tree.nodes, cr.merkle(leaves, threads=0))
    rng = np.random.default_rng(7)
    idx = rng.integers(0, m, size=k).astype(np.uint64)
    paths = np.stack([tree.path(int(i)) for i in idx])
    lv = leaves[idx.astype(np.int64)].copy()
    expect = np.ones(k, dtype=bool)
    for j in range(0, k, 7):                           # a corrupted sibling somewhere on the path
        paths[j, int(rng.integers(0, depth)), int(rng.integers(0, 3))] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 60)))
        expect[j] = False
    for j in range(3, k, 11):                          # somebody else's leaf
        lv[j] = leaves[(int(idx[j]) + 1) % m]
        expect[j] = False
    for j in range(5, k, 13):                          # right walk, but the index claims a position outside the tree
        idx[j] += np.uint64(m) << np.uint64(int(rng.integers(0, 8)))
        expect[j] = False
    got = S.verify_paths(cfg, lv, idx, paths, tree.root)
    assert np.array_equal(got, expect)
    # the oracle's walk for a sample, bad ones included (left / right by the index bits below `depth`)
    for j in list(range(0, 64)) + [k - 1]:
        cur = lv[j:j + 1].copy()
        for level in range(depth):
            pair = np.zeros((1, 2, 4), dtype=np.uint64)
            right = (int(idx[j]) >> level) & 1
            pair[0, right], pair[0, 1 - right] = cur[0], paths[j, level]
            cur = cr.hash_batch(pair, 2, 1, threads=1).reshape(1, 4)
        assert bool(np.array_equal(cur[0], tree.root) and int(idx[j]) < m) == bool(got[j]), j


def test_pinned_host_buffers_take_the_pipelined_path_and_agree():
    """Page-locked host buffers (pmx_host_alloc) switch pmx_permute_batch / pmx_hash_batch to the chunked three-stream pipeline (upload,
    kernel and download of different chunks at once); pageable buffers of 16 MiB and more are page-locked for the length of the call and
    take the same pipeline.  Results must equal each other and the oracle, also for batch sizes that do not divide evenly (the 2^18
    states here are 24 MiB: the registered path)."""
    import ctypes
    from sponge_amd import _lib
    cfg = product_config("bls_t3_a5_8_31")
    ctx = cfg.context()
    for n in (100, (1 << 16) + 77, 1 << 17, (1 << 18) + 5):
        states = synth.random_elements(cfg.field, n * 3, seed=n).reshape(n, 3, 4)
        want = ctx.permute_batch(states)                       # pageable (page-locked by the call from 16 MiB)
        pin = S.pinned_empty((n, 3, 4))
        pin[:] = states
        ctx.permute_batch_inplace(pin)
        assert np.array_equal(pin, want), n
    n = (1 << 16) + 5
    msgs = synth.random_elements(cfg.field, n * 4, seed=3).reshape(n, 4, 4)
    want = ctx.hash_batch(msgs, 4, 2)
    pin_in, pin_out = S.pinned_empty((n, 4, 4)), S.pinned_empty((n, 2, 4))
    pin_in[:] = msgs
    _lib.check(_lib.lib().pmx_hash_batch(ctx._h, ctypes.c_void_p(pin_in.ctypes.data), 4, ctypes.c_void_p(pin_out.ctypes.data), 2, n))
    assert np.array_equal(pin_out, want)
    sample = np.arange(0, n, 997)
    assert np.array_equal(want[sample], c_oracle("bls_t3_a5_8_31").hash_batch(np.ascontiguousarray(msgs[sample]), 4, 2, threads=0))


@pytest.mark.parametrize("rate,capacity", [(1, 2), (3, 0), (2, 1), (4, 5)])
def test_other_rate_capacity_splits(rate, capacity):
    """PoseidonConfig::new accepts any rate/capacity split of the width (mod.rs:187-213); the default tables only use
    capacity 1.  Same constants, different split: absorb / squeeze against the Python oracle."""
    from oracle import poseidon_oracle as O
    f = S.BLS12_381_FR
    t = rate + capacity
    base = S.poseidon_config_from_lfsr(f, t - 1, 5, 8, 20)           # constants for this width
    cfg = S.PoseidonConfig(f, 8, 20, 5, base.mds, base.ark, rate, capacity)
    ob = O.make_config(O.BLS12_381_FR, 255, t - 1, 5, 8, 20)
    ocfg = O.PoseidonConfig(ob.p, 8, 20, 5, ob.ark, ob.mds, rate, capacity)
    msg = [7, 8, 9, 10, 11]
    sponge, osp = S.PoseidonSponge.new(cfg), O.PoseidonSponge(ocfg)
    sponge.absorb(f.from_ints(msg))
    osp.absorb(msg)
    assert f.to_ints(sponge.squeeze_native_field_elements(2 * rate + 1)) == osp.squeeze_native_field_elements(2 * rate + 1)
    sponge.absorb(f.from_ints(msg[:2]))
    osp.absorb(msg[:2])
    assert f.to_ints(sponge.squeeze_native_field_elements(1)) == osp.squeeze_native_field_elements(1)
    assert f.to_ints(sponge.state) == osp.state and (sponge.mode.tag, sponge.mode.index) == (osp.mode, osp.index)


def test_merkle_tree_with_rate3_capacity0_crosses_the_narrow_level_threshold():
    """Width 3 as (rate 3, capacity 0): 2-to-1 compression is permute([l, r, 0])[0] there, which the quad kernel of
    the narrow levels (state [0, l, r], lane 1; capacity 1 only) does not compute - every level of this split has to
    take the one-lane-per-state kernel.  A 2^16-leaf tree lies entirely below the 32768-compression switch to the quad kernel; all nodes against the
    C restatement, and the batched path verifier against the tree."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.BLS12_381_FR
    base = S.poseidon_config_from_lfsr(f, 2, 5, 8, 31)
    cfg = S.PoseidonConfig(f, 8, 31, 5, base.mds, base.ark, 3, 0)
    ob = O.make_config(O.BLS12_381_FR, 255, 2, 5, 8, 31)
    cr = cref.CRef(O.PoseidonConfig(ob.p, 8, 31, 5, ob.ark, ob.mds, 3, 0))
    m = 1 << 16
    leaves = synth.random_elements(f, m, seed=0x5EED0030)
    nodes, root = cfg.context().merkle_2to1(leaves)
    want = cr.merkle(leaves, threads=0)
    assert np.array_equal(nodes, want) and np.array_equal(root, want[-1])
    tree = S.MerkleTree(cfg, leaves)
    idx = [0, 1, 12345, m - 1]
    assert S.verify_paths(cfg, leaves[idx], idx, [tree.path(i) for i in idx], tree.root).all()


@pytest.mark.parametrize("rf,rp", [(300, 8), (120, 31), (600, 0)])
def test_configs_whose_round_constants_exceed_lds_fall_back_to_the_run_time_engine(rf, rp):
    """The t = 3 register engine stages 144 B of round constants per round in LDS (576 B in the quad kernel): hundreds
    of rounds do not fit a workgroup's LDS.  The reference handles any round count (mod.rs:95-118), so such configs run
    on the engine that reads its constants through the scalar cache - every entry point, against the C restatement."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.BLS12_381_FR
    cfg = S.poseidon_config_from_lfsr(f, 2, 5, rf, rp)
    cr = cref.CRef(O.make_config(O.BLS12_381_FR, 255, 2, 5, rf, rp))
    ctx = cfg.context()
    n = 300
    states = synth.random_elements(f, n * 3, seed=rf).reshape(n, 3, 4)
    assert np.array_equal(ctx.permute_batch(states), cr.permute_batch(states, threads=0))
    msgs = synth.random_elements(f, n * 5, seed=rf + 1).reshape(n, 5, 4)
    assert np.array_equal(ctx.hash_batch(msgs, 5, 3), cr.hash_batch(msgs, 5, 3, threads=0))
    leaves = synth.random_elements(f, 256, seed=rf + 2)
    nodes, _ = ctx.merkle_2to1(leaves)
    assert np.array_equal(nodes, cr.merkle(leaves, threads=0))


def _random_configs():
    import random
    rng = random.Random(0xC0FFEE)
    out = []
    for _ in range(24):
        field = rng.choice(["bls12_381_fr", "bn254_fr"])
        rate = rng.randint(1, 11)
        alpha = rng.choice([3, 5, 5, 7, 17, 257])
        rf = rng.choice([2, 4, 8])
        rp = rng.choice([0, 1, 2, 7, 22, 31, 57, 60, 70])
        out.append((field, rate, alpha, rf, rp))
    return out


@pytest.mark.parametrize("field_name,rate,alpha,rf,rp", _random_configs())
def test_random_configs_vs_c_oracle(field_name, rate, alpha, rf, rp):
    """Two dozen seeded random (field, width, exponent, round counts): whatever engine the dispatch picks - register,
    hybrid, run-time width, dense or optimised schedule, partial sections of 0 / 1 / 2 rounds, sections too long for the
    uncapped identity lanes - permutation, hash driver and a small tree agree with the C restatement built from the
    ORACLE's own constants."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.FIELDS[field_name]
    p, bits = {"bls12_381_fr": (O.BLS12_381_FR, 255), "bn254_fr": (O.BN254_FR, 254)}[field_name]
    cfg = S.poseidon_config_from_lfsr(f, rate, alpha, rf, rp)
    cr = cref.CRef(O.make_config(p, bits, rate, alpha, rf, rp))
    ctx, t = cfg.context(), rate + 1
    n = 131
    states = synth.random_elements(f, n * t, seed=rate * 1000 + rp).reshape(n, t, 4)
    states[0] = 0
    states[1] = f.from_ints([p - 1] * t)
    assert np.array_equal(ctx.permute_batch(states), cr.permute_batch(states, threads=0))
    L = t + 2
    msgs = synth.random_elements(f, n * L, seed=rate * 1000 + rp + 1).reshape(n, L, 4)
    assert np.array_equal(ctx.hash_batch(msgs, L, 2), cr.hash_batch(msgs, L, 2, threads=0))
    if rate >= 2:
        leaves = synth.random_elements(f, 64, seed=rate + rp)
        nodes, _ = ctx.merkle_2to1(leaves)
        assert np.array_equal(nodes, cr.merkle(leaves, threads=0))


def test_device_resident_flow_with_the_abis_own_memory_helpers():
    """pmx_device_alloc / _upload / _download / pmx_stream_synchronize: the *_dev entry points driven without torch or
    any HIP binding on the caller's side - permute twice in HBM, hash the result in HBM, download."""
    import ctypes
    from sponge_amd import _lib
    lib = _lib.lib()
    cfg = product_config("bls_t3_a5_8_31")
    ctx = cfg.context()
    n = 70000
    states = synth.random_elements(cfg.field, n * 3, seed=0x5EED0050).reshape(n, 3, 4)
    d_states, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.pmx_device_alloc(0, ctypes.byref(d_states), states.nbytes))
    _lib.check(lib.pmx_device_alloc(0, ctypes.byref(d_out), n * 32))
    _lib.check(lib.pmx_device_upload(0, d_states, ctypes.c_void_p(states.ctypes.data), states.nbytes, None))
    ctx.permute_batch_dev(d_states.value, n, 0)
    ctx.permute_batch_dev(d_states.value, n, 0)
    ctx.hash_batch_dev(d_states.value, 3, d_out.value, 1, n, 0)            # every permuted state as a 3-element message
    back, digests = np.zeros_like(states), np.zeros((n, 1, 4), dtype=np.uint64)
    _lib.check(lib.pmx_device_download(0, ctypes.c_void_p(back.ctypes.data), d_states, back.nbytes, None))
    _lib.check(lib.pmx_device_download(0, ctypes.c_void_p(digests.ctypes.data), d_out, digests.nbytes, None))
    _lib.check(lib.pmx_stream_synchronize(0, None))
    cr = c_oracle("bls_t3_a5_8_31")
    want = cr.permute_batch(cr.permute_batch(states, threads=0), threads=0)
    assert np.array_equal(back, want)
    assert np.array_equal(digests, cr.hash_batch(want, 3, 1, threads=0))
    _lib.check(lib.pmx_device_free(0, d_states))
    _lib.check(lib.pmx_device_free(0, d_out))
    assert lib.pmx_device_alloc(99, ctypes.byref(d_states), 16) == _lib.PMX_ERR_ARG


@pytest.mark.parametrize("n", [32768, 32769])
def test_sponge_driver_at_the_quad_kernel_switch(n):
    """Up to 32768 mid-stream t = 3 sponges run on the quad kernels (one state per four lanes), more on the one-lane
    kernels: both sides of the switch, mixed modes and indices, against the C restatement."""
    name = "bls_t3_a5_8_31"
    cfg = product_config(name)
    cr = c_oracle(name)
    f = cfg.field
    b = S.BatchPoseidonSponge.new(cfg, n)
    b.state[:] = synth.random_elements(f, n * 3, seed=n).reshape(n, 3, 4)
    b.mode_index[::2] = 1
    b.mode_index[5::7] = 2
    b.mode_tag[3::5] = 1                                  # PMX_MODE_SQUEEZING, indices 0 / 1 / 2 mixed in
    st0, tag0, idx0 = b.state.copy(), b.mode_tag.copy(), b.mode_index.copy()
    msgs = synth.random_elements(f, n * 3, seed=n + 1).reshape(n, 3, 4)
    b.absorb(msgs)
    got = b.squeeze_native_field_elements(4)
    for i in list(range(0, 64)) + list(range(n - 64, n)) + list(range(64, n - 64, 97)):
        s, m, x = cr.sponge_absorb(st0[i], int(tag0[i]), int(idx0[i]), msgs[i])
        s, m, x, out = cr.sponge_squeeze(s, m, x, 4)
        assert np.array_equal(got[i], out) and np.array_equal(b.state[i], s), i
        assert (int(b.mode_tag[i]), int(b.mode_index[i])) == (m, x), i


@pytest.mark.parametrize("alpha", [5, 17, 3])
def test_t3_window_engine_on_a_modulus_whose_residues_exceed_32_balanced_bytes(alpha):
    """2^255 - 19 at t = 3: calls above the quad kernels' range run on the window engine like every other modulus (round 6: residues
    above what 32 balanced bytes reach are stored as Y - p; until then this modulus had a register engine of its own).  40000 units and
    2^17 + 77 through permute, the hash driver and a tree, whole batches against the C port."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    import ctypes
    p = (1 << 255) - 19
    f = S.Field("p25519", p)
    cfg = S.poseidon_config_from_lfsr(f, 2, alpha, 8, 31)
    cr = cref.CRef(O.make_config(p, 255, 2, alpha, 8, 31))
    ctx = cfg.context()
    for n in (40000, (1 << 17) + 77):
        info = _lib.PmxEngineInfo()
        _lib.check(_lib.lib().pmx_ctx_engine_info(ctx._h, _lib.OP_PERMUTE, n, 0, ctypes.byref(info)))
        assert info.engine.startswith(b"HybridEngine<3,") and info.mfma_dense == 1 and info.partial_window == 3, info.engine
        states = synth.random_elements(f, n * 3, seed=0x5EED0060 + alpha).reshape(n, 3, 4)
        assert np.array_equal(ctx.permute_batch(states), cr.permute_batch(states, threads=0)), n
        msgs = synth.random_elements(f, n * 5, seed=0x5EED0061 + alpha).reshape(n, 5, 4)
        assert np.array_equal(ctx.hash_batch(msgs, 5, 3), cr.hash_batch(msgs, 5, 3, threads=0)), n
    leaves = synth.random_elements(f, 1 << 19, seed=alpha)          # levels of 2^18 ... 2^16 compressions on the window engine, the rest on the quad kernels
    nodes, _ = ctx.merkle_2to1(leaves)
    assert np.array_equal(nodes, cr.merkle(leaves, threads=0))


@pytest.mark.parametrize("name", ["bls_t3_a5_8_31", "bls_t3_a17_8_31"])
def test_t3_calls_just_above_the_quad_range(name):
    """t = 3 launches of 32769 units and more run on the window engine (below: the quad kernels): 40000 units - less than one wave per
    SIMD - through permute and the hash driver, whole batch against the C port."""
    cfg = product_config(name)
    cr = c_oracle(name)
    n = 40000
    states = synth.random_elements(cfg.field, n * 3, seed=0x5EED0060).reshape(n, 3, 4)
    assert np.array_equal(cfg.context().permute_batch(states), cr.permute_batch(states, threads=0))
    msgs = synth.random_elements(cfg.field, n * 5, seed=0x5EED0061).reshape(n, 5, 4)
    assert np.array_equal(cfg.context().hash_batch(msgs, 5, 3), cr.hash_batch(msgs, 5, 3, threads=0))


@pytest.mark.parametrize("n_trees,m", [(1, 64), (3, 16), (5, 1), (1000, 4), (77, 128), (40000, 2), (300, 1024)])
def test_merkle_forest_vs_c_oracle(n_trees, m):
    """pmx_merkle_2to1_forest: n_trees trees of m leaves advanced together level by level (a level of the forest is one launch
    over all trees).  Every node of every tree against the C port's tree of that tree's leaves; the level-major layout of the
    header; a forest of one tree is pmx_merkle_2to1's node array.  A parent is new; absorb([l, r]); squeeze_native(1)
    (mod.rs:219-254, 321-341)."""
    name = "bls_t3_a5_8_31"
    cfg = product_config(name)
    cr = c_oracle(name)
    leaves = synth.random_elements(cfg.field, n_trees * m, seed=0x5EED00F0 + n_trees + m).reshape(n_trees, m, 4)
    nodes, roots = cfg.context().merkle_2to1_forest(leaves, n_trees)
    assert nodes.shape == (n_trees * (2 * m - 1), 4)
    # the C port builds ONE tree over all the leaves: the forest is its first log2(m) levels, and those ARE level-major
    # (pairs never straddle two trees); pad the tree count to a power of two for the checker only
    pad = 1 << (n_trees - 1).bit_length()
    padded = np.zeros((pad * m, 4), dtype=np.uint64)
    padded[:n_trees * m] = leaves.reshape(-1, 4)
    big = cr.merkle(padded, threads=0)
    src_f = src_b = 0
    width = m
    while width >= 1:
        assert np.array_equal(nodes[src_f:src_f + n_trees * width], big[src_b:src_b + n_trees * width]), (n_trees, m, width)
        src_f += n_trees * width
        src_b += pad * width
        width //= 2
    assert np.array_equal(roots, nodes[-n_trees:])
    for b in sorted({0, n_trees // 2, n_trees - 1}):                       # and a few trees on their own
        assert np.array_equal(roots[b], cr.merkle(leaves[b], threads=1)[-1])
    if n_trees == 1:
        one, root = cfg.context().merkle_2to1(leaves[0])
        assert np.array_equal(one, nodes) and np.array_equal(root, roots[0])


def test_merkle_forest_bad_arguments():
    cfg = product_config("bls_t3_a5_8_31")
    leaves = synth.random_elements(cfg.field, 12, seed=1)
    with pytest.raises(S.PmxError):
        cfg.context().merkle_2to1_forest(leaves, 4)                          # 3 leaves per tree: not a power of two
    lib = S.lib()
    import ctypes
    assert lib.pmx_merkle_2to1_forest(cfg.context()._h, ctypes.c_void_p(leaves.ctypes.data), 0, 4, None, None) == -2
    assert lib.pmx_merkle_2to1_forest(cfg.context()._h, None, 1, 4, None, None) == -2
    assert lib.pmx_merkle_2to1_forest_dev(cfg.context()._h, None, 1, 4, None) == -2


@pytest.mark.parametrize("rate", [2, 3, 5, 8, 9])
@pytest.mark.parametrize("alpha", [0, 1, 2])
def test_small_exponents_on_every_engine(rate, alpha):
    """The optimised schedules add an S-box output into rows unreduced, and pmx_mfma.hpp cuts it into 32 bytes - one k-step of the
    matrix-core instruction per element: both stand on the bound of a Montgomery PRODUCT.  alpha = 1 used to hand its lazy input
    through (wrong results at t = 6 ... 9) and is formed as the product x * 1 since (pmx_field.hpp: fe_sbox), so it runs the same
    engines as every other exponent.  Whole permutations at both ends of the engine thresholds and a hash, against the C port (the
    reference accepts any alpha: src/poseidon/mod.rs:63-74)."""
    import ctypes
    from oracle import cref
    from oracle import poseidon_oracle as O
    f, t = S.BLS12_381_FR, rate + 1
    cfg = S.poseidon_config_from_lfsr(f, rate, alpha, 8, 57)
    cr = cref.CRef(O.make_config(O.BLS12_381_FR, 255, rate, alpha, 8, 57))
    info = _lib.PmxEngineInfo()
    _lib.check(_lib.lib().pmx_ctx_engine_info(cfg.context()._h, _lib.OP_PERMUTE, 1 << 18, 0, ctypes.byref(info)))
    assert info.mfma_dense == (1 if t <= 9 else 0), info.engine
    assert (b"mfma" in info.engine) == (t <= 9), info.engine
    pm = f.modulus
    for n in (333, (1 << 17) + 5) if t == 3 else (333,):
        states = synth.random_elements(f, n * t, seed=900 + 10 * rate + alpha).reshape(n, t, 4)
        states[0] = f.from_ints([pm - 1] * t).reshape(t, 4)
        states[1] = f.from_ints([0] * t).reshape(t, 4)
        assert np.array_equal(cfg.context().permute_batch(states), cr.permute_batch(states, threads=0)), n
    msgs = synth.random_elements(f, 70 * (rate + 2), seed=alpha).reshape(70, rate + 2, 4)
    assert np.array_equal(cfg.context().hash_batch(msgs, rate + 2, 2), cr.hash_batch(msgs, rate + 2, 2, threads=0))
