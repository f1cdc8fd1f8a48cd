"""Absorb / AbsorbWithLength encodings of the host mirror (sponge_amd/absorb.py, src/absorb.rs) against the oracle's
separate restatement, plus the reference's own tests of this layer re-expressed (src/poseidon/tests.rs:26-117)."""
import random

import numpy as np
import pytest

import sponge_amd as S
from sponge_amd import absorb as A
from oracle import absorb_oracle as AO
from oracle import poseidon_oracle as O

F = S.BLS12_381_FR
P, BITS = O.BLS12_381_FR, 255


def pairs():
    rng = random.Random(3)
    rb = bytes(rng.randrange(256) for _ in range(100))
    return [
        (A.U8(200), ("u8", 200)), (A.U16(65535), ("u16", 65535)), (A.U32(7), ("u32", 7)),
        (A.U64(2**64 - 1), ("u64", 2**64 - 1)), (A.U128(2**127 + 5), ("u128", 2**127 + 5)), (A.Usize(12), ("usize", 12)),
        (A.I8(-128), ("i8", -128)), (A.I16(-2), ("i16", -2)), (A.I32(123456), ("i32", 123456)),
        (A.I64(-2**63), ("i64", -2**63)), (A.I128(-1), ("i128", -1)), (A.Isize(-77), ("isize", -77)),
        (A.Bool(True), ("bool", True)), (A.Bool(False), ("bool", False)),
        (A.Fp(P - 1, F), ("fp", P - 1)), (A.Fp(5, F), ("fp", 5)),
        (A.Bytes(b""), ("vec", "u8", [])), (A.Bytes(b"abc"), ("vec", "u8", list(b"abc"))),
        (A.Bytes(rb[:23]), ("vec", "u8", list(rb[:23]))), (A.Bytes(rb[:31]), ("vec", "u8", list(rb[:31]))),
        (A.Bytes(rb), ("vec", "u8", list(rb))),
        (A.Seq([A.I32(v) for v in (1, 2, 3, 4, 5, 6)]), ("vec", "i32", [1, 2, 3, 4, 5, 6])),
        (A.Seq([A.Fp(v, F) for v in (9, 8, P - 2)]), ("vec", "fp", [9, 8, P - 2])),
        (A.Seq([A.Bytes(b"xy"), A.Bytes(b"")], A.Seq), ("vec", "vec", [("vec", "u8", list(b"xy")), ("vec", "u8", [])])),
        (A.Opt(None), ("opt", None)), (A.Opt(A.U8(3)), ("opt", ("u8", 3))),
        (A.WithLength(A.Bytes(bytes([1, 2, 3, 4]))), ("with_len", ("vec", "u8", [1, 2, 3, 4]))),
        (A.WithLength(A.Seq([A.U64(5), A.U64(6)])), ("with_len", ("vec", "u64", [5, 6]))),
        (A.TEAffine(11, P - 3, F), ("te", 11, P - 3)), (A.SWAffine(5, 6, False, F), ("sw", 5, 6, False)),
        (A.SWAffine(0, 1, True, F), ("sw", 0, 1, True)),
        (A.Seq([A.TEAffine(1, 2, F), A.TEAffine(3, 4, F)]), ("vec", "te", [("te", 1, 2), ("te", 3, 4)])),
    ]


@pytest.mark.parametrize("obj,spec", pairs())
def test_encodings_match_oracle(obj, spec):
    assert obj.to_sponge_field_elements_as_vec(F) == AO.field_elements(P, BITS, spec)
    assert obj.to_sponge_bytes_as_vec() == AO.sponge_bytes(BITS, spec)


def test_byte_packing_boundaries():
    # 8-byte length prefix + payload, 31 usable bytes per BLS12-381 / BN254 element (src/absorb.rs:135-139)
    for n, want in [(0, 1), (23, 1), (24, 2), (54, 2), (55, 3)]:
        assert len(A.Bytes(bytes(n)).to_sponge_field_elements_as_vec(F)) == want
    assert all(v < 2**248 for v in A.Bytes(bytes([255]) * 200).to_sponge_field_elements_as_vec(F))
    assert len(A.Bytes(bytes(54)).to_sponge_field_elements_as_vec(S.BN254_FR)) == 2


def test_curve_points():
    """src/absorb.rs:232-254: a point absorbs as its base-field coordinates ([x, y] / [x, y, infinity]); a point over
    another field panics (field_cast(..).unwrap()); the byte form carries the Vec's u64 element count."""
    te, sw = A.TEAffine(7, 9, F), A.SWAffine(7, 9, False, F)
    assert te.to_sponge_field_elements_as_vec(F) == [7, 9]
    assert sw.to_sponge_field_elements_as_vec(F) == [7, 9, 0]
    assert A.SWAffine(0, 1, True, F).to_sponge_field_elements_as_vec(F)[2] == 1
    assert len(te.to_sponge_bytes_as_vec()) == 8 + 2 * 32 and te.to_sponge_bytes_as_vec()[:8] == (2).to_bytes(8, "little")
    assert len(sw.to_sponge_bytes_as_vec()) == 8 + 3 * 32
    with pytest.raises(ValueError):
        A.TEAffine(7, 9, S.BN254_FR).to_sponge_field_elements_as_vec(F)


def test_macros():
    """src/poseidon/tests.rs:87-117 (the parts that need no sponge)."""
    expected = bytearray()
    A.Seq([A.I32(v) for v in (6, 5, 4, 3, 2, 1)]).to_sponge_bytes(expected)
    A.Fp(42, F).to_sponge_bytes(expected)
    assert A.collect_sponge_bytes(A.Seq([A.I32(v) for v in (6, 5, 4, 3, 2, 1)]), A.Fp(42, F)) == bytes(expected)
    exp = []
    A.Seq([A.I32(v) for v in (6, 5, 4, 3, 2, 1)]).to_sponge_field_elements(F, exp)
    A.Fp(42, F).to_sponge_field_elements(F, exp)
    assert A.collect_sponge_field_elements(F, A.Seq([A.I32(v) for v in (6, 5, 4, 3, 2, 1)]), A.Fp(42, F)) == exp


def test_variable_size_lists_have_different_encodings():
    """src/poseidon/tests.rs:57-69: [[1,2,3,4],[5,6]] vs [[1,2],[3,4,5,6]] with per-list lengths."""
    lst1 = A.Seq([A.WithLength(A.Bytes(bytes([1, 2, 3, 4]))), A.WithLength(A.Bytes(bytes([5, 6])))], A.WithLength)
    lst2 = A.Seq([A.WithLength(A.Bytes(bytes([1, 2]))), A.WithLength(A.Bytes(bytes([3, 4, 5, 6])))], A.WithLength)
    assert lst1.to_sponge_bytes_as_vec() != lst2.to_sponge_bytes_as_vec()
    assert lst1.to_sponge_field_elements_as_vec(F) != lst2.to_sponge_field_elements_as_vec(F)
    # without the lengths the byte encodings collide - the reason AbsorbWithLength exists
    flat1 = A.Seq([A.Bytes(bytes([1, 2, 3, 4])), A.Bytes(bytes([5, 6]))], A.Seq)
    flat2 = A.Seq([A.Bytes(bytes([1, 2])), A.Bytes(bytes([3, 4, 5, 6]))], A.Seq)
    assert flat1.to_sponge_bytes_as_vec() == flat2.to_sponge_bytes_as_vec()


def test_non_native_field_elements():
    other = A.Fp(5, S.BN254_FR)
    assert other.to_sponge_field_elements_as_vec(F) == []          # `let _ = field_cast(..)`, absorb.rs:157
    with pytest.raises(ValueError):
        A.Seq([other]).to_sponge_field_elements_as_vec(F)          # field_cast(batch).unwrap(), absorb.rs:163


@pytest.mark.gpu
def test_absorb_objects_fork_and_nonnative_squeeze_on_gpu():
    """absorb(&impl Absorb), the absorb! macro order (tests.rs:87-99), fork (src/lib.rs:149-157) and the non-native
    squeeze (src/lib.rs:61-100) through the GPU sponge, against the oracle sponge fed by the oracle's encodings."""
    from gpu_helpers import product_config
    from helpers import oracle_config
    cfg = product_config("reference_test_a17_8_29")
    ocfg = oracle_config("reference_test_a17_8_29")
    sponge1 = S.PoseidonSponge.new(cfg)
    sponge1.absorb(A.Seq([A.I32(v) for v in (1, 2, 3, 4, 5, 6)]))
    sponge1.absorb(A.Fp(114514, F))
    osp = O.PoseidonSponge(ocfg)
    osp.absorb(AO.field_elements(P, BITS, ("vec", "i32", [1, 2, 3, 4, 5, 6])))
    osp.absorb(AO.field_elements(P, BITS, ("fp", 114514)))
    forked = sponge1.fork(b"domain-separator")
    oforked = osp.clone()
    oforked.absorb(AO.field_elements(P, BITS, AO.fork_input(b"domain-separator")))
    assert F.to_ints(sponge1.squeeze_native_field_elements(3)) == osp.squeeze_native_field_elements(3)
    assert F.to_ints(forked.squeeze_native_field_elements(3)) == oforked.squeeze_native_field_elements(3)
    # non-native squeeze into BN254 Fr: 253 bits per element, little-endian, reduced mod p2
    got = forked.squeeze_field_elements(4, S.BN254_FR)
    bits = oforked.squeeze_bits(253 * 4, 255)
    want = [sum(b << i for i, b in enumerate(bits[k * 253:(k + 1) * 253])) % O.BN254_FR for k in range(4)]
    assert got == want
    # native squeeze with sizes (src/lib.rs:166-182): all Full == plain squeeze; a Truncated entry switches to the
    # bit recomposition, where every element still takes MODULUS_BIT_SIZE - 1 = 254 bits (src/lib.rs:45-52)
    s1, s2, o1 = forked.clone(), forked.clone(), oforked.clone()
    assert s1.squeeze_native_field_elements_with_sizes([None, None]) == o1.squeeze_native_field_elements(2)
    obits = oforked.clone().squeeze_bits(254 * 2, 255)
    assert s2.squeeze_native_field_elements_with_sizes([None, 100]) == \
        [sum(b << i for i, b in enumerate(obits[k * 254:(k + 1) * 254])) % P for k in range(2)]
    with pytest.raises(ValueError):
        forked.clone().squeeze_native_field_elements_with_sizes([256])      # panic at src/lib.rs:48
    # squeeze_field_elements_with_sizes::<F2> with F2 of the native characteristic (mod.rs:288-299) is the native
    # squeeze: all-Full sizes return whole native elements, not 254-bit truncations
    s3, s4 = forked.clone(), forked.clone()
    assert s3.squeeze_field_elements_with_sizes([None, None], F) == oforked.clone().squeeze_native_field_elements(2)
    assert s4.squeeze_field_elements_with_sizes([None, 100], F) == \
        [sum(b << i for i, b in enumerate(obits[k * 254:(k + 1) * 254])) % P for k in range(2)]
    # single_field_element (tests.rs:26-33): elem and elem + 1 give different outputs
    a, b = S.PoseidonSponge.new(cfg), S.PoseidonSponge.new(cfg)
    a.absorb(A.Fp(987654321, F))
    b.absorb(A.Fp(987654322, F))
    assert not np.array_equal(a.squeeze_native_field_elements(3), b.squeeze_native_field_elements(3))
