"""CPU oracle for the Poseidon hot path of arkworks-rs/sponge  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-Python big-integer restatement of the reference's algorithm.  It is the
*checker* for the HIP product path; nothing under ``sponge_amd/`` may import it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use ``oracle/``.

Parity pin: the reference is Rust and cannot be built in this image (no cargo/rustc, ark-ff is an
un-vendored git dependency, Cargo.toml:36-42).  The oracle is therefore pinned against every
known-answer value the reference's own tests hold for this path (see ``oracle/kats.py`` and
``tests/test_oracle_kats.py``):
  * ``test_grain_lfsr_consistency``                         src/poseidon/grain_lfsr.rs:197-213
  * ``bls12_381_fr_poseidon_default_parameters_test``       src/poseidon/traits.rs:163-358 (28 values)
  * ``test_poseidon_sponge_consistency``                    src/poseidon/mod.rs:376-399 (3 outputs)

All values here are *canonical* integers in [0, p).  ``to_mont``/``from_mont`` convert to the
boundary representation (Montgomery residue x*2^256 mod p as 4 little-endian u64 limbs, which is how
ark-ff's ``Fp<MontBackend<_,4>,4>`` stores an element).

Reference lines followed, function by function:
  GrainLFSR                 src/poseidon/grain_lfsr.rs:15-189
  find_poseidon_ark_and_mds src/poseidon/traits.rs:105-146
  default table             src/test.rs:13-32, src/poseidon/traits.rs:69-102
  permute / apply_*         src/poseidon/mod.rs:63-118
  absorb / squeeze          src/poseidon/mod.rs:121-182, 232-254, 321-341
  squeeze_bytes / bits      src/poseidon/mod.rs:256-286
  DuplexSpongeMode          src/lib.rs:198-210
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

# ----------------------------------------------------------------------------------------------
# Fields.  BLS12-381 Fr is the reference's only test field (src/test.rs:6); BN254 Fr does not
# occur in the reference and is carried for BASELINE.json config C3.
# ----------------------------------------------------------------------------------------------
BLS12_381_FR = 52435875175126190479447740508185965837690552500527637822603658699938581184513
BN254_FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617

MONT_BITS = 256
MONT_R = 1 << MONT_BITS
LIMB_MASK = (1 << 64) - 1

# (rate, alpha, full_rounds, partial_rounds, skip_matrices)   src/test.rs:14-31
BLS12_381_FR_OPT_CONSTRAINTS = [
    (2, 17, 8, 31, 0), (3, 5, 8, 56, 0), (4, 5, 8, 56, 0), (5, 5, 8, 57, 0),
    (6, 5, 8, 57, 0), (7, 5, 8, 57, 0), (8, 5, 8, 57, 0),
]
BLS12_381_FR_OPT_WEIGHTS = [(r, 257, 8, 13, 0) for r in range(2, 9)]


def to_limbs(x: int) -> List[int]:
    return [(x >> (64 * i)) & LIMB_MASK for i in range(4)]


def from_limbs(limbs: Sequence[int]) -> int:
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs))


def to_mont(x: int, p: int) -> int:
    return (x * MONT_R) % p


def from_mont(x: int, p: int) -> int:
    return (x * pow(MONT_R, -1, p)) % p


def mont_constants(p: int) -> dict:
    """inv = -p^-1 mod 2^64, R mod p, R^2 mod p  (the numbers SURVEY.md section 8c tabulates)."""
    return {
        "inv": (-pow(p, -1, 1 << 64)) % (1 << 64),
        "r": MONT_R % p,
        "r2": (MONT_R * MONT_R) % p,
    }


# ----------------------------------------------------------------------------------------------
# Grain LFSR (grain_lfsr.rs).  The 80-bit register is kept as a Python list used as a ring; the
# reference keeps [bool; 80] + head, the tap positions are relative to head (grain_lfsr.rs:163-168).
# ----------------------------------------------------------------------------------------------
class GrainLFSR:
    TAPS = (62, 51, 38, 23, 13, 0)

    def __init__(self, is_sbox_inverse: bool, prime_bits: int, t: int, rf: int, rp: int):
        self.prime_bits = prime_bits
        bits = [0] * 80
        bits[1] = 1                                   # b0 b1 = 0 1 : prime field  (grain_lfsr.rs:25)
        bits[5] = 1 if is_sbox_inverse else 0         # b2..b5 S-box flag           (:28-32)

        def put(lo: int, hi: int, value: int) -> None:  # big-endian value into bits[lo..hi]
            for pos in range(hi, lo - 1, -1):
                bits[pos] = value & 1
                value >>= 1

        put(6, 17, prime_bits)                        # :35-41
        put(18, 29, t)                                # :44-50
        put(30, 39, rf)                               # :53-59
        put(40, 49, rp)                               # :62-68
        for pos in range(50, 80):                     # :71-73
            bits[pos] = 1
        self.bits = bits
        self.head = 0
        for _ in range(160):                          # warm-up, :176-188
            self._clock()

    def _clock(self) -> int:
        h = self.head
        b = 0
        for tap in self.TAPS:
            b ^= self.bits[(h + tap) % 80]
        self.bits[h] = b
        self.head = (h + 1) % 80
        return b

    def next_bit(self) -> int:
        """One output bit of the self-shrinking generator (grain_lfsr.rs:89-103)."""
        while True:
            first = self._clock()
            second = self._clock()
            if first:
                return second

    def next_int(self) -> int:
        """prime_bits bits, first bit most significant (grain_lfsr.rs:119-123, 141-153)."""
        v = 0
        for _ in range(self.prime_bits):
            v = (v << 1) | self.next_bit()
        return v

    def field_elements_rejection(self, n: int, p: int) -> List[int]:
        out = []
        while len(out) < n:                           # :115-129
            v = self.next_int()
            if v < p:
                out.append(v)
        return out

    def field_elements_mod_p(self, n: int, p: int) -> List[int]:
        return [self.next_int() % p for _ in range(n)]  # :139-156


def find_poseidon_ark_and_mds(p: int, prime_bits: int, rate: int, rf: int, rp: int,
                              skip_matrices: int) -> Tuple[List[List[int]], List[List[int]]]:
    """traits.rs:105-146.  ark[round][i], mds[i][j] = (xs[i] + ys[j])^-1, canonical integers."""
    t = rate + 1
    lfsr = GrainLFSR(False, prime_bits, t, rf, rp)
    ark = [lfsr.field_elements_rejection(t, p) for _ in range(rf + rp)]
    for _ in range(skip_matrices):
        lfsr.field_elements_mod_p(2 * t, p)
    xs = lfsr.field_elements_mod_p(t, p)
    ys = lfsr.field_elements_mod_p(t, p)
    mds = [[pow((xs[i] + ys[j]) % p, -1, p) for j in range(t)] for i in range(t)]
    return ark, mds


# ----------------------------------------------------------------------------------------------
# Config (mod.rs:23-42, 185-214)
# ----------------------------------------------------------------------------------------------
@dataclass
class PoseidonConfig:
    p: int
    full_rounds: int
    partial_rounds: int
    alpha: int
    ark: List[List[int]]
    mds: List[List[int]]
    rate: int
    capacity: int

    def __post_init__(self):
        t = self.rate + self.capacity                 # the asserts of PoseidonConfig::new, mod.rs:196-203
        assert len(self.ark) == self.full_rounds + self.partial_rounds
        assert all(len(r) == t for r in self.ark)
        assert len(self.mds) == t and all(len(r) == t for r in self.mds)

    @property
    def t(self) -> int:
        return self.rate + self.capacity


def default_bls12_381_config(rate: int, optimized_for_weights: bool) -> PoseidonConfig:
    """Fr::get_default_poseidon_parameters (traits.rs:69-102) for the table of src/test.rs."""
    table = BLS12_381_FR_OPT_WEIGHTS if optimized_for_weights else BLS12_381_FR_OPT_CONSTRAINTS
    for (r, alpha, rf, rp, skip) in table:
        if r == rate:
            ark, mds = find_poseidon_ark_and_mds(BLS12_381_FR, 255, rate, rf, rp, skip)
            return PoseidonConfig(BLS12_381_FR, rf, rp, alpha, ark, mds, rate, 1)  # capacity 1, :96
    raise KeyError(rate)


def make_config(p: int, prime_bits: int, rate: int, alpha: int, rf: int, rp: int,
                skip: int = 0) -> PoseidonConfig:
    """Any (field, rate, alpha, RF, RP): LFSR constants with capacity 1 (alpha is not in the seed)."""
    ark, mds = find_poseidon_ark_and_mds(p, prime_bits, rate, rf, rp, skip)
    return PoseidonConfig(p, rf, rp, alpha, ark, mds, rate, 1)


# ----------------------------------------------------------------------------------------------
# Permutation (mod.rs:63-118): every round is ARK -> S-box -> MDS.
# ----------------------------------------------------------------------------------------------
def permute(cfg: PoseidonConfig, state: Sequence[int]) -> List[int]:
    p, t, alpha = cfg.p, cfg.t, cfg.alpha
    s = list(state)
    half = cfg.full_rounds // 2
    total = cfg.full_rounds + cfg.partial_rounds
    for rnd in range(total):
        ark = cfg.ark[rnd]
        s = [(s[i] + ark[i]) % p for i in range(t)]                         # apply_ark   :76-80
        if rnd < half or rnd >= half + cfg.partial_rounds:                 # full round  :65-69
            s = [pow(x, alpha, p) for x in s]
        else:                                                              # partial     :71-73
            s[0] = pow(s[0], alpha, p)
        s = [sum(cfg.mds[i][j] * s[j] for j in range(t)) % p for i in range(t)]  # apply_mds :82-93
    return s


# ----------------------------------------------------------------------------------------------
# Duplex sponge (mod.rs:121-182, 216-254, 320-342; lib.rs:198-210)
# ----------------------------------------------------------------------------------------------
ABSORBING = 0
SQUEEZING = 1


@dataclass
class PoseidonSponge:
    cfg: PoseidonConfig
    state: List[int] = field(default_factory=list)
    mode: int = ABSORBING        # DuplexSpongeMode tag
    index: int = 0               # next_absorb_index / next_squeeze_index

    def __post_init__(self):
        if not self.state:
            self.state = [0] * self.cfg.t             # CryptographicSponge::new, mod.rs:219-230

    def clone(self) -> "PoseidonSponge":
        return PoseidonSponge(self.cfg, list(self.state), self.mode, self.index)

    def _permute(self) -> None:
        self.state = permute(self.cfg, self.state)

    def _absorb_internal(self, start: int, elems: Sequence[int]) -> None:     # mod.rs:121-150
        p, rate, cap = self.cfg.p, self.cfg.rate, self.cfg.capacity
        rem = list(elems)
        while True:
            if start + len(rem) <= rate:
                for i, e in enumerate(rem):
                    k = cap + start + i
                    self.state[k] = (self.state[k] + e) % p
                self.mode, self.index = ABSORBING, start + len(rem)
                return
            take = rate - start
            for i in range(take):
                k = cap + start + i
                self.state[k] = (self.state[k] + rem[i]) % p
            self._permute()
            rem = rem[take:]
            start = 0

    def _squeeze_internal(self, start: int, n: int) -> List[int]:             # mod.rs:153-182
        rate, cap = self.cfg.rate, self.cfg.capacity
        out: List[int] = []
        remaining = n
        while True:
            if start + remaining <= rate:
                out.extend(self.state[cap + start: cap + start + remaining])
                self.mode, self.index = SQUEEZING, start + remaining
                return out
            take = rate - start
            out.extend(self.state[cap + start: cap + start + take])
            if remaining != rate:              # mod.rs:175 -- tested BEFORE the slice is advanced
                self._permute()
            remaining -= take
            start = 0

    def absorb(self, elems: Sequence[int]) -> None:                           # mod.rs:232-254
        if len(elems) == 0:
            return
        if self.mode == ABSORBING:
            idx = self.index
            if idx == self.cfg.rate:
                self._permute()
                idx = 0
            self._absorb_internal(idx, elems)
        else:
            self._permute()
            self._absorb_internal(0, elems)

    def squeeze_native_field_elements(self, n: int) -> List[int]:             # mod.rs:321-341
        if self.mode == ABSORBING:
            self._permute()
            return self._squeeze_internal(0, n)
        idx = self.index
        if idx == self.cfg.rate:
            self._permute()
            idx = 0
        return self._squeeze_internal(idx, n)

    def squeeze_bytes(self, num_bytes: int, prime_bits: int) -> bytes:        # mod.rs:256-270
        usable = (prime_bits - 1) // 8
        n = (num_bytes + usable - 1) // usable
        out = bytearray()
        for e in self.squeeze_native_field_elements(n):
            out += e.to_bytes(32, "little")[:usable]
        return bytes(out[:num_bytes])

    def squeeze_bits(self, num_bits: int, prime_bits: int) -> List[int]:      # mod.rs:272-286
        usable = prime_bits - 1
        n = (num_bits + usable - 1) // usable
        out: List[int] = []
        for e in self.squeeze_native_field_elements(n):
            out.extend((e >> k) & 1 for k in range(usable))
        return out[:num_bits]


def hash_fixed(cfg: PoseidonConfig, msg: Sequence[int], n_out: int) -> List[int]:
    """new; absorb(msg); squeeze_native(n_out) -- the batch driver's per-row contract."""
    sp = PoseidonSponge(cfg)
    sp.absorb(msg)
    return sp.squeeze_native_field_elements(n_out)


def compress_2to1(cfg: PoseidonConfig, left: int, right: int) -> int:
    """Merkle 2-to-1: new; absorb([l, r]); squeeze 1  ==  permute([0, l, r])[capacity] for rate>=2."""
    return hash_fixed(cfg, [left, right], 1)[0]


def merkle_levels(cfg: PoseidonConfig, leaves: Sequence[int]) -> List[List[int]]:
    """All levels, leaves first, root last.  len(leaves) must be a power of two."""
    n = len(leaves)
    assert n and (n & (n - 1)) == 0
    levels = [list(leaves)]
    while len(levels[-1]) > 1:
        cur = levels[-1]
        levels.append([compress_2to1(cfg, cur[2 * i], cur[2 * i + 1]) for i in range(len(cur) // 2)])
    return levels
