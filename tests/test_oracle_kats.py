"""Pins the Python oracle against every known-answer value the reference's tests hold."""
import pytest

from oracle import kats as K
from oracle import poseidon_oracle as O

from helpers import golden, ints, oracle_config


def test_grain_lfsr_consistency():
    # src/poseidon/grain_lfsr.rs:197-213
    lfsr = O.GrainLFSR(False, 255, 3, 8, 31)
    a = lfsr.field_elements_rejection(1, O.BLS12_381_FR)[0]
    b = lfsr.field_elements_rejection(1, O.BLS12_381_FR)[0]
    assert [a, b] == K.GRAIN_LFSR_255_3_8_31


@pytest.mark.parametrize("rate,weights", sorted(K.DEFAULT_PARAMS_BLS12_381))
def test_bls12_381_fr_poseidon_default_parameters(rate, weights):
    # src/poseidon/traits.rs:163-358
    cfg = O.default_bls12_381_config(rate, weights)
    ark00, mds00 = K.DEFAULT_PARAMS_BLS12_381[(rate, weights)]
    assert cfg.ark[0][0] == ark00
    assert cfg.mds[0][0] == mds00
    assert cfg.capacity == 1 and cfg.rate == rate


def test_poseidon_sponge_consistency():
    # src/poseidon/mod.rs:376-399
    cfg = O.default_bls12_381_config(2, False)
    sponge = O.PoseidonSponge(cfg)
    sponge.absorb(K.SPONGE_CONSISTENCY_INPUT)
    assert sponge.squeeze_native_field_elements(3) == K.SPONGE_CONSISTENCY_OUTPUT


def test_montgomery_constants_match_survey_table():
    # SURVEY.md section 8c table
    c = O.mont_constants(O.BLS12_381_FR)
    assert c["inv"] == 0xFFFFFFFEFFFFFFFF
    assert O.to_limbs(c["r"]) == [0x1FFFFFFFE, 0x5884B7FA00034802, 0x998C4FEFECBC4FF5, 0x1824B159ACC5056F]
    assert O.to_limbs(c["r2"]) == [0xC999E990F3F29C6D, 0x2B6CEDCB87925C23, 0x05D314967254398F, 0x0748D9D99F59FF11]
    c = O.mont_constants(O.BN254_FR)
    assert c["inv"] == 0xC2E1F593EFFFFFFF
    assert O.to_limbs(c["r"]) == [0xAC96341C4FFFFFFB, 0x36FC76959F60CD29, 0x666EA36F7879462E, 0x0E0A77C19A07DF2F]


def test_golden_permute_vectors_reproduce():
    for name, vecs in golden("permute_vectors.json").items():
        cfg = oracle_config(name)
        for v in vecs[:4]:
            assert O.permute(cfg, ints(v["in"])) == ints(v["out"]), name


def test_golden_traces_reproduce_and_cover_quirk():
    traces = golden("sponge_traces.json")
    for name, by_trace in traces.items():
        cfg = oracle_config(name)
        for tname, steps in by_trace.items():
            sp = O.PoseidonSponge(cfg)
            for st in steps:
                if st["op"] == "absorb":
                    sp.absorb(ints(st["in"]))
                else:
                    assert sp.squeeze_native_field_elements(st["n"]) == ints(st["out"]), (name, tname)
                assert sp.state == ints(st["state"]) and [sp.mode, sp.index] == st["mode"]
    # mod.rs:175: Squeezing{1} + squeeze(rate): second output comes from the SAME un-permuted state
    q = traces["bls_t3_a5_8_31"]["quirk_175"]
    assert q[1]["mode"] == [O.SQUEEZING, 1]
    assert q[2]["state"] == q[1]["state"] and q[2]["mode"] == [O.SQUEEZING, 1]
    assert ints(q[2]["out"]) == [ints(q[1]["state"])[2], ints(q[1]["state"])[1]]


def test_compress_equals_permute_of_zero_capacity():
    cfg = oracle_config("bls_t3_a5_8_31")
    assert O.compress_2to1(cfg, 1, 2) == O.permute(cfg, [0, 1, 2])[1]


def test_squeeze_bytes_and_bits_lengths():
    # src/poseidon/mod.rs:256-286: 31 usable bytes / 254 usable bits per BLS12-381 element
    cfg = oracle_config("bls_t3_a17_8_31")
    sp = O.PoseidonSponge(cfg)
    sp.absorb([1, 2, 3])
    a, b = sp.clone(), sp.clone()
    by = a.squeeze_bytes(70, 255)
    bits = b.squeeze_bits(300, 255)
    elems = sp.squeeze_native_field_elements(3)
    assert len(by) == 70 and len(bits) == 300
    assert by[:31] == elems[0].to_bytes(32, "little")[:31]
    assert by[62:70] == elems[2].to_bytes(32, "little")[:8]
    assert bits[254:300] == [(elems[1] >> k) & 1 for k in range(46)]
