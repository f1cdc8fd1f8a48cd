"""Prime fields at the ABI boundary: elements are [..., 4] uint64 arrays of Montgomery limbs
(R = 2^256), the in-memory form of ark-ff's Fp<MontBackend<_,4>,4>.  Conversions run through the
library's host-side pmx_to_mont / pmx_from_mont."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List

import numpy as np

from . import _lib

_MASK = (1 << 64) - 1


def _limbs(x: int) -> List[int]:
    return [(x >> (64 * i)) & _MASK for i in range(4)]


@dataclass(frozen=True)
class Field:
    name: str
    modulus: int

    @property
    def modulus_bit_size(self) -> int:          # PrimeField::MODULUS_BIT_SIZE
        return self.modulus.bit_length()

    def modulus_limbs(self) -> np.ndarray:
        return np.array(_limbs(self.modulus), dtype=np.uint64)

    def from_ints(self, values: Iterable[int]) -> np.ndarray:
        """canonical integers -> [n][4] Montgomery limbs (Fr::from / MontFp!)."""
        vals = [int(v) % self.modulus for v in values]
        arr = np.array([_limbs(v) for v in vals], dtype=np.uint64).reshape(len(vals), 4)
        if len(vals):
            m = self.modulus_limbs()
            _lib.check(_lib.lib().pmx_to_mont(m.ctypes.data, arr.ctypes.data, len(vals)))
        return arr

    def to_ints(self, elems: np.ndarray) -> List[int]:
        """[..., 4] Montgomery limbs -> canonical integers (into_bigint)."""
        arr = np.ascontiguousarray(elems, dtype=np.uint64).reshape(-1, 4).copy()
        if arr.shape[0]:
            m = self.modulus_limbs()
            _lib.check(_lib.lib().pmx_from_mont(m.ctypes.data, arr.ctypes.data, arr.shape[0]))
        return [sum(int(row[i]) << (64 * i) for i in range(4)) for row in arr]

    def mont_constants(self):
        m = self.modulus_limbs()
        inv = np.zeros(1, dtype=np.uint64)
        r = np.zeros(4, dtype=np.uint64)
        r2 = np.zeros(4, dtype=np.uint64)
        _lib.check(_lib.lib().pmx_mont_constants(m.ctypes.data, inv.ctypes.data, r.ctypes.data, r2.ctypes.data))
        return int(inv[0]), r, r2


# src/test.rs:6 -- the reference's test field
BLS12_381_FR = Field("bls12_381_fr",
                     52435875175126190479447740508185965837690552500527637822603658699938581184513)
# not in the reference; BASELINE.json config C3
BN254_FR = Field("bn254_fr",
                 21888242871839275222246405745257275088548364400416034343698204186575808495617)

FIELDS = {f.name: f for f in (BLS12_381_FR, BN254_FR)}
