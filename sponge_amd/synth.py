"""Synthetic field-element batches for the benchmark and the full-size property tests.

Counter-based: limb j of element i is splitmix64(seed + 4*i + j); the top limb is masked to the bit size
of the modulus and p is subtracted once if the value is >= p (2^bits < 2p for both fields).  Any reduced
value is a valid Montgomery residue, so the batch is used as-is."""
from __future__ import annotations

import numpy as np

from .field import Field

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(counter: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (counter.astype(np.uint64) + np.uint64(1)) * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def random_elements(field: Field, n: int, seed: int, offset: int = 0) -> np.ndarray:
    """[n][4] uint64, every element in [0, p).  `offset` = index of the first element (for sharding)."""
    with np.errstate(over="ignore"):
        ctr = np.uint64(seed) + np.arange(4 * offset, 4 * (offset + n), dtype=np.uint64)
    limbs = splitmix64(ctr).reshape(n, 4)
    bits = field.modulus_bit_size
    top_bits = bits - 192
    limbs[:, 3] &= np.uint64((1 << top_bits) - 1)
    p = field.modulus_limbs()
    # lexicographic compare from the top limb: ge = (limbs >= p)
    ge = np.ones(n, dtype=bool)
    decided = np.zeros(n, dtype=bool)
    for i in (3, 2, 1, 0):
        gt = limbs[:, i] > p[i]
        lt = limbs[:, i] < p[i]
        ge = np.where(~decided & lt, False, ge)
        decided |= gt | lt
    # subtract p where needed (multi-limb borrow)
    borrow = np.zeros(n, dtype=np.uint64)
    out = limbs.copy()
    with np.errstate(over="ignore"):
        for i in range(4):
            a = limbs[:, i]
            d = a - p[i] - borrow
            nb = ((a < p[i]) | ((a == p[i]) & (borrow == 1))).astype(np.uint64)
            out[:, i] = np.where(ge, d, a)
            borrow = nb
    return out
