"""The C restatement (oracle/poseidon_ref.c) against the KAT-pinned Python oracle, limb for limb."""
import random

import numpy as np
import pytest

from oracle import cref
from oracle import kats as K
from oracle import poseidon_oracle as O

from helpers import golden, ints, oracle_config

CONFIGS = ["bls_t3_a5_8_31", "bls_t3_a17_8_31", "bls_t4_a5_8_56", "bls_t9_a5_8_57",
           "bls_t3_a257_8_13", "bn254_t9_a5_8_57", "bn254_t3_a5_8_57", "reference_test_a17_8_29"]


@pytest.mark.parametrize("name", CONFIGS)
def test_c_permute_matches_golden(name):
    cfg = oracle_config(name)
    cr = cref.CRef(cfg)
    vecs = golden("permute_vectors.json")[name]
    flat = [x for v in vecs for x in ints(v["in"])]
    states = cref.elems_to_limbs(flat, cfg.p).reshape(len(vecs), cfg.t, 4)
    out = cr.permute_batch(states, threads=2)
    got = cref.limbs_to_elems(out, cfg.p)
    want = [x for v in vecs for x in ints(v["out"])]
    assert got == want
    # outputs are fully reduced residues
    assert all(O.from_limbs([int(x) for x in row]) < cfg.p for row in out.reshape(-1, 4))


def test_c_reference_kat_through_sponge_calls():
    # src/poseidon/mod.rs:376-399 through the C mode machine
    cfg = O.default_bls12_381_config(2, False)
    cr = cref.CRef(cfg)
    state = np.zeros((3, 4), dtype=np.uint64)
    state, mode, idx = cr.sponge_absorb(state, O.ABSORBING, 0,
                                        cref.elems_to_limbs(K.SPONGE_CONSISTENCY_INPUT, cfg.p))
    state, mode, idx, out = cr.sponge_squeeze(state, mode, idx, 3)
    assert cref.limbs_to_elems(out, cfg.p) == K.SPONGE_CONSISTENCY_OUTPUT
    assert (mode, idx) == (O.SQUEEZING, 1)


@pytest.mark.parametrize("name", ["bls_t3_a5_8_31", "reference_test_a17_8_29", "bn254_t9_a5_8_57"])
def test_c_sponge_traces(name):
    cfg = oracle_config(name)
    cr = cref.CRef(cfg)
    for tname, steps in golden("sponge_traces.json")[name].items():
        state, mode, idx = np.zeros((cfg.t, 4), dtype=np.uint64), O.ABSORBING, 0
        for st in steps:
            if st["op"] == "absorb":
                state, mode, idx = cr.sponge_absorb(state, mode, idx,
                                                    cref.elems_to_limbs(ints(st["in"]), cfg.p))
            else:
                state, mode, idx, out = cr.sponge_squeeze(state, mode, idx, st["n"])
                assert cref.limbs_to_elems(out, cfg.p) == ints(st["out"]), (tname,)
            assert cref.limbs_to_elems(state, cfg.p) == ints(st["state"]), (tname,)
            assert [mode, idx] == st["mode"], (tname,)


def test_c_hash_and_merkle_golden():
    g = golden("hash_merkle_vectors.json")
    for name, d in g.items():
        cfg = oracle_config(name)
        cr = cref.CRef(cfg)
        for row in d["hash"]:
            msg = cref.elems_to_limbs(ints(row["in"]), cfg.p).reshape(1, row["L"], 4)
            out = cr.hash_batch(msg, row["L"], row["k"])
            assert cref.limbs_to_elems(out, cfg.p) == ints(row["out"])
    cfg = oracle_config("bls_t3_a5_8_31")
    levels = g["bls_t3_a5_8_31"]["merkle16"]
    nodes = cref.CRef(cfg).merkle(cref.elems_to_limbs(ints(levels[0]), cfg.p), threads=2)
    assert cref.limbs_to_elems(nodes, cfg.p) == [x for lvl in levels for x in ints(lvl)]


def test_c_random_batch_vs_python_and_thread_invariance():
    cfg = oracle_config("bls_t3_a5_8_31")
    cr = cref.CRef(cfg)
    rng = random.Random(7)
    vals = [rng.randrange(cfg.p) for _ in range(64 * 3)]
    states = cref.elems_to_limbs(vals, cfg.p).reshape(64, 3, 4)
    a = cr.permute_batch(states, threads=1)
    b = cr.permute_batch(states, threads=4)
    assert np.array_equal(a, b)
    want = [x for i in range(64) for x in O.permute(cfg, vals[3 * i:3 * i + 3])]
    assert cref.limbs_to_elems(a, cfg.p) == want
