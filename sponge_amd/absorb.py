"""Input encodings of the reference's `Absorb` / `AbsorbWithLength` traits (src/absorb.rs) for the host mirror.

Python has no u8/i32/... types, so values are wrapped: U8(7), I32(-3), Bool(True), Bytes(b"..") (= &[u8] / Vec<u8>),
Fp(x) (a native field element given as a canonical integer), Seq([...]) (= &[A] / Vec<A>), Opt(None | value),
WithLength(Seq|Bytes) (= to_sponge_*_with_length).  Every wrapper offers the two methods of the trait:

    to_sponge_bytes(dest: bytearray)                         src/absorb.rs:16
    to_sponge_field_elements(field, dest: list[int])         src/absorb.rs:28   (canonical integers)

Encodings follow src/absorb.rs line by line; two of them lean on ark-ff (not in the reference tree):
  * `[u8]::to_field_elements()` (ark-ff `ToConstraintField<F> for [u8]`): chunks of (MODULUS_BIT_SIZE-1)/8 bytes,
    each read as a little-endian integer (always < p) - used for byte slices after the u64-LE length prefix
    (src/absorb.rs:135-139);
  * `Fp::serialize_compressed` (ark-serialize): ceil(MODULUS_BIT_SIZE/8) bytes, little-endian canonical value
    (src/absorb.rs:152-154);
  * curve points (src/absorb.rs:232-254) go through ark-ec's `ToConstraintField`: a twisted-Edwards affine point is
    [x, y], a short-Weierstrass affine point [x, y, infinity as 0/1] over the base field; their byte form is that
    Vec<BaseField> under `serialize_compressed` (ark-serialize: u64-LE element count, then every element).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

from .field import Field


def bytes_to_field_elements(field: Field, data: bytes) -> List[int]:
    """ark-ff ToConstraintField<F> for [u8]."""
    max_size = (field.modulus_bit_size - 1) // 8
    return [int.from_bytes(data[i:i + max_size], "little") for i in range(0, len(data), max_size)]


class Absorb:
    def to_sponge_bytes(self, dest: bytearray) -> None:
        raise NotImplementedError

    def to_sponge_field_elements(self, field: Field, dest: List[int]) -> None:
        raise NotImplementedError

    def to_sponge_bytes_as_vec(self) -> bytes:
        d = bytearray()
        self.to_sponge_bytes(d)
        return bytes(d)

    def to_sponge_field_elements_as_vec(self, field: Field) -> List[int]:
        d: List[int] = []
        self.to_sponge_field_elements(field, d)
        return d

    # batch_* defaults (src/absorb.rs:40-78): element by element
    @classmethod
    def batch_to_sponge_bytes(cls, batch, dest: bytearray) -> None:
        for a in batch:
            a.to_sponge_bytes(dest)

    @classmethod
    def batch_to_sponge_field_elements(cls, field: Field, batch, dest: List[int]) -> None:
        for a in batch:
            a.to_sponge_field_elements(field, dest)


class _Unsigned(Absorb):
    BYTES = 0

    def __init__(self, v: int):
        assert 0 <= v < (1 << (8 * self.BYTES)), "value out of range"
        self.v = v

    def to_sponge_bytes(self, dest):                       # src/absorb.rs:169-171
        dest += self.v.to_bytes(self.BYTES, "little")

    def to_sponge_field_elements(self, field, dest):       # F::from(x), :173-175
        dest.append(self.v % field.modulus)


class U8(_Unsigned):
    BYTES = 1

    @classmethod
    def batch_to_sponge_field_elements(cls, field, batch, dest):   # src/absorb.rs:135-139
        data = bytes(a.v for a in batch)
        dest.extend(bytes_to_field_elements(field, len(data).to_bytes(8, "little") + data))


class U16(_Unsigned):
    BYTES = 2


class U32(_Unsigned):
    BYTES = 4


class U64(_Unsigned):
    BYTES = 8


class U128(_Unsigned):
    BYTES = 16


class Usize(U64):                                          # src/absorb.rs:212-220: as u64
    pass


class _Signed(Absorb):
    BYTES = 0

    def __init__(self, v: int):
        assert -(1 << (8 * self.BYTES - 1)) <= v < (1 << (8 * self.BYTES - 1)), "value out of range"
        self.v = v

    def to_sponge_bytes(self, dest):                       # two's complement LE, :186-188
        dest += self.v.to_bytes(self.BYTES, "little", signed=True)

    def to_sponge_field_elements(self, field, dest):       # +-F::from(|x|), :190-196
        dest.append(self.v % field.modulus)


class I8(_Signed):
    BYTES = 1


class I16(_Signed):
    BYTES = 2


class I32(_Signed):
    BYTES = 4


class I64(_Signed):
    BYTES = 8


class I128(_Signed):
    BYTES = 16


class Isize(I64):                                          # src/absorb.rs:222-230
    pass


class Bool(Absorb):                                        # src/absorb.rs:142-150
    def __init__(self, v: bool):
        self.v = bool(v)

    def to_sponge_bytes(self, dest):
        dest.append(1 if self.v else 0)

    def to_sponge_field_elements(self, field, dest):
        dest.append(1 if self.v else 0)


class Fp(Absorb):
    """A field element (canonical integer) of `of_field`; native when of_field is the sponge's field."""

    def __init__(self, value: int, of_field: Field):
        self.v = value % of_field.modulus
        self.f = of_field

    def to_sponge_bytes(self, dest):                       # serialize_compressed, :153-155
        dest += self.v.to_bytes((self.f.modulus_bit_size + 7) // 8, "little")

    def to_sponge_field_elements(self, field, dest):       # field_cast, :156-158 (`let _ =`: non-native is a no-op)
        if field.modulus == self.f.modulus:
            dest.append(self.v)

    @classmethod
    def batch_to_sponge_field_elements(cls, field, batch, dest):   # field_cast(batch).unwrap(), :159-164
        for a in batch:
            if a.f.modulus != field.modulus:
                raise ValueError("Trying to absorb non-native field elements (field_cast returned None)")
            dest.append(a.v)


class _AffinePoint(Absorb):
    """Shared by TEAffine / SWAffine (src/absorb.rs:232-254): `self.to_field_elements()` over the curve's base field,
    then - bytes: that vector under serialize_compressed; field elements: field_cast::<BaseField, F>(..).unwrap(),
    i.e. absorbing a point whose base field is not the sponge's field panics."""

    def __init__(self, base_field: Field):
        self.base = base_field

    def to_field_elements(self) -> List[int]:
        raise NotImplementedError

    def to_sponge_bytes(self, dest):
        elems = self.to_field_elements()
        dest += len(elems).to_bytes(8, "little")           # Vec<T>::serialize: the length as u64
        for v in elems:
            dest += v.to_bytes((self.base.modulus_bit_size + 7) // 8, "little")

    def to_sponge_field_elements(self, field, dest):
        if field.modulus != self.base.modulus:
            raise ValueError("Trying to absorb non-native field elements (field_cast returned None)")
        dest.extend(self.to_field_elements())


class TEAffine(_AffinePoint):
    """Twisted-Edwards affine point (x, y), canonical integers of `base_field` (src/absorb.rs:232-242)."""

    def __init__(self, x: int, y: int, base_field: Field):
        super().__init__(base_field)
        self.x, self.y = x % base_field.modulus, y % base_field.modulus

    def to_field_elements(self):
        return [self.x, self.y]


class SWAffine(_AffinePoint):
    """Short-Weierstrass affine point (x, y, infinity) (src/absorb.rs:244-254)."""

    def __init__(self, x: int, y: int, infinity: bool, base_field: Field):
        super().__init__(base_field)
        self.x, self.y, self.infinity = x % base_field.modulus, y % base_field.modulus, bool(infinity)

    def to_field_elements(self):
        return [self.x, self.y, 1 if self.infinity else 0]


class Seq(Absorb):
    """&[A] / Vec<A> of one element type (src/absorb.rs:256-288): the element type's batch encoders."""

    def __init__(self, items: Iterable[Absorb], elem_type: Optional[type] = None):
        self.items = list(items)
        self.elem_type = elem_type or (type(self.items[0]) if self.items else Absorb)

    def to_sponge_bytes(self, dest):
        self.elem_type.batch_to_sponge_bytes(self.items, dest)

    def to_sponge_field_elements(self, field, dest):
        self.elem_type.batch_to_sponge_field_elements(field, self.items, dest)

    def absorb_length(self) -> int:
        return len(self.items)


def Bytes(data: bytes) -> Seq:
    """A byte string as &[u8]."""
    return Seq([U8(b) for b in data], U8)


class WithLength(Absorb):
    """AbsorbWithLength::to_sponge_*_with_length (src/absorb.rs:84-101): the usize length, then the contents."""

    def __init__(self, seq: Seq):
        self.seq = seq

    def to_sponge_bytes(self, dest):
        Usize(self.seq.absorb_length()).to_sponge_bytes(dest)
        self.seq.to_sponge_bytes(dest)

    def to_sponge_field_elements(self, field, dest):
        Usize(self.seq.absorb_length()).to_sponge_field_elements(field, dest)
        self.seq.to_sponge_field_elements(field, dest)


class Opt(Absorb):                                         # src/absorb.rs:290-304
    def __init__(self, item: Optional[Absorb]):
        self.item = item

    def to_sponge_bytes(self, dest):
        Bool(self.item is not None).to_sponge_bytes(dest)
        if self.item is not None:
            self.item.to_sponge_bytes(dest)

    def to_sponge_field_elements(self, field, dest):
        Bool(self.item is not None).to_sponge_field_elements(field, dest)
        if self.item is not None:
            self.item.to_sponge_field_elements(field, dest)


def collect_sponge_bytes(*items: Absorb) -> bytes:         # macro, src/absorb.rs:331-341
    d = bytearray()
    for it in items:
        it.to_sponge_bytes(d)
    return bytes(d)


def collect_sponge_field_elements(field: Field, *items: Absorb) -> List[int]:   # macro, :345-355
    d: List[int] = []
    for it in items:
        it.to_sponge_field_elements(field, d)
    return d
