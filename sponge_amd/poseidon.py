"""Host-side mirror of the reference's Poseidon interface over the C ABI.

Names, argument meaning and behaviour follow the reference (file:line in /root/reference):
  PoseidonConfig / PoseidonConfig.new            src/poseidon/mod.rs:23-42, 185-214
  PoseidonSponge (parameters, state, mode)       src/poseidon/mod.rs:51-60
    .new / .absorb / .squeeze_bytes / .squeeze_bits / .squeeze_field_elements   :216-318
    .squeeze_native_field_elements               :320-342
    .into_state / .from_state                    :344-367 (SpongeExt, src/lib.rs:188-195)
  DuplexSpongeMode                               src/lib.rs:198-210
  find_poseidon_ark_and_mds                      src/poseidon/traits.rs:105-146
  get_default_poseidon_parameters                src/poseidon/traits.rs:59-102 with the table of src/test.rs:13-32
Every permutation runs on the GPU through libposeidon_mi355x.so; a single PoseidonSponge is the n = 1
case of the batch driver (BASELINE config C1, "plumbing").  BatchPoseidonSponge is the same interface
over n independent sponges.

Field elements are numpy uint64 arrays [..., 4] of Montgomery limbs (see field.py).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import MODE_ABSORBING, MODE_SQUEEZING, PmxError
from .field import BLS12_381_FR, Field


# ---------------------------------------------------------------------------------------------------
# DuplexSpongeMode
# ---------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class DuplexSpongeMode:
    tag: int      # MODE_ABSORBING / MODE_SQUEEZING
    index: int    # next_absorb_index / next_squeeze_index

    @staticmethod
    def Absorbing(next_absorb_index: int) -> "DuplexSpongeMode":
        return DuplexSpongeMode(MODE_ABSORBING, next_absorb_index)

    @staticmethod
    def Squeezing(next_squeeze_index: int) -> "DuplexSpongeMode":
        return DuplexSpongeMode(MODE_SQUEEZING, next_squeeze_index)


# ---------------------------------------------------------------------------------------------------
# PoseidonConfig
# ---------------------------------------------------------------------------------------------------
class PoseidonConfig:
    """ark: [full_rounds+partial_rounds][t][4], mds: [t][t][4] Montgomery limbs, mds[i][j] row-major."""

    def __init__(self, field: Field, full_rounds: int, partial_rounds: int, alpha: int, mds: np.ndarray,
                 ark: np.ndarray, rate: int, capacity: int):
        t = rate + capacity
        mds = np.ascontiguousarray(mds, dtype=np.uint64)
        ark = np.ascontiguousarray(ark, dtype=np.uint64)
        # the asserts of PoseidonConfig::new, src/poseidon/mod.rs:196-203
        assert ark.shape == (full_rounds + partial_rounds, t, 4), "ark must be [RF+RP][rate+capacity]"
        assert mds.shape == (t, t, 4), "mds must be [rate+capacity][rate+capacity]"
        self.field = field
        self.full_rounds = full_rounds
        self.partial_rounds = partial_rounds
        self.alpha = alpha
        self.mds = mds
        self.ark = ark
        self.rate = rate
        self.capacity = capacity
        self._ctx: dict = {}

    new = classmethod(lambda cls, *a, **k: cls(*a, **k))

    @property
    def t(self) -> int:
        return self.rate + self.capacity

    def context(self, device: int = 0) -> "Context":
        """The device context holding this config's constants (created once per device)."""
        if device not in self._ctx:
            self._ctx[device] = Context(self, device)
        return self._ctx[device]


def find_poseidon_ark_and_mds(field: Field, prime_bits: int, rate: int, full_rounds: int, partial_rounds: int,
                              skip_matrices: int) -> Tuple[np.ndarray, np.ndarray]:
    """src/poseidon/traits.rs:105-146 (Grain LFSR + Cauchy matrix), run by the library's host code."""
    t = rate + 1
    ark = np.zeros((full_rounds + partial_rounds, t, 4), dtype=np.uint64)
    mds = np.zeros((t, t, 4), dtype=np.uint64)
    m = field.modulus_limbs()
    _lib.check(_lib.lib().pmx_find_poseidon_ark_and_mds(m.ctypes.data, prime_bits, rate, full_rounds, partial_rounds,
                                                        skip_matrices, ark.ctypes.data, mds.ctypes.data))
    return ark, mds


# PoseidonDefaultConfig for BLS12-381 Fr: (rate, alpha, full_rounds, partial_rounds, skip_matrices), src/test.rs:13-32
PARAMS_OPT_FOR_CONSTRAINTS = {
    BLS12_381_FR.name: [(2, 17, 8, 31, 0), (3, 5, 8, 56, 0), (4, 5, 8, 56, 0), (5, 5, 8, 57, 0), (6, 5, 8, 57, 0),
                        (7, 5, 8, 57, 0), (8, 5, 8, 57, 0)],
}
PARAMS_OPT_FOR_WEIGHTS = {
    BLS12_381_FR.name: [(r, 257, 8, 13, 0) for r in range(2, 9)],
}


def get_default_poseidon_parameters(field: Field, rate: int, optimized_for_weights: bool) -> Optional[PoseidonConfig]:
    """PoseidonDefaultConfigField::get_default_poseidon_parameters (src/poseidon/traits.rs:69-102)."""
    table = (PARAMS_OPT_FOR_WEIGHTS if optimized_for_weights else PARAMS_OPT_FOR_CONSTRAINTS).get(field.name)
    if table is None:
        return None
    for (r, alpha, rf, rp, skip) in table:
        if r == rate:
            ark, mds = find_poseidon_ark_and_mds(field, field.modulus_bit_size, rate, rf, rp, skip)
            return PoseidonConfig(field, rf, rp, alpha, mds, ark, rate, 1)  # capacity 1, traits.rs:96
    return None


def poseidon_config_from_lfsr(field: Field, rate: int, alpha: int, full_rounds: int, partial_rounds: int,
                              skip_matrices: int = 0) -> PoseidonConfig:
    """LFSR constants for an arbitrary (field, rate, alpha, RF, RP); capacity 1.  alpha is not part of the
    Grain seed, so e.g. BASELINE's (t=3, alpha=5, 8, 31) shares its constants with the default rate-2 entry."""
    ark, mds = find_poseidon_ark_and_mds(field, field.modulus_bit_size, rate, full_rounds, partial_rounds,
                                         skip_matrices)
    return PoseidonConfig(field, full_rounds, partial_rounds, alpha, mds, ark, rate, 1)


# ---------------------------------------------------------------------------------------------------
# Pinned host memory
# ---------------------------------------------------------------------------------------------------
class _PinnedOwner:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            _lib.lib().pmx_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(shape) -> np.ndarray:
    """uint64 array in page-locked host memory (pmx_host_alloc): the host-buffer entry points then pipeline the
    PCIe copies with the kernel.  The memory lives as long as the returned array (or views of it)."""
    n = int(np.prod(shape))
    ptr = ctypes.c_void_p()
    _lib.check(_lib.lib().pmx_host_alloc(ctypes.byref(ptr), max(n, 1) * 8))
    buf = (ctypes.c_uint64 * max(n, 1)).from_address(ptr.value)
    buf._pmx_owner = _PinnedOwner(ptr)          # freed when the last view of the buffer goes away
    return np.frombuffer(buf, dtype=np.uint64, count=n).reshape(shape)


# ---------------------------------------------------------------------------------------------------
# Device context
# ---------------------------------------------------------------------------------------------------
def _ptr(a: Optional[np.ndarray]):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def c_config(cfg: "PoseidonConfig") -> "_lib.PmxConfig":
    """pmx_config view of a PoseidonConfig (borrows cfg.ark / cfg.mds: keep `cfg` alive while it is used)."""
    c = _lib.PmxConfig()
    c.full_rounds, c.partial_rounds, c.alpha = cfg.full_rounds, cfg.partial_rounds, cfg.alpha
    c.rate, c.capacity = cfg.rate, cfg.capacity
    for i, l in enumerate(cfg.field.modulus_limbs()):
        c.modulus[i] = int(l)
    c.ark = cfg.ark.ctypes.data
    c.mds = cfg.mds.ctypes.data
    return c


class Context:
    """pmx_ctx: one validated config resident on one GPU, taken from the library's process-wide cache
    (pmx_ctx_acquire): equal configs share one device context however many PoseidonConfig objects carry them."""

    def __init__(self, cfg: PoseidonConfig, device: int = 0):
        self.cfg = cfg
        self.device = device
        c = c_config(cfg)
        handle = ctypes.c_void_p()
        _lib.check(_lib.lib().pmx_ctx_acquire(ctypes.byref(c), device, ctypes.byref(handle)))
        self._h = handle

    @classmethod
    def borrowed(cls, cfg: PoseidonConfig, device: int, handle: int) -> "Context":
        """A view of a context somebody else owns (a device group's): never released from here."""
        self = cls.__new__(cls)
        self.cfg, self.device, self._h, self._borrowed = cfg, device, ctypes.c_void_p(handle), True
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                _lib.lib().pmx_ctx_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer entry points ------------------------------------------------------------
    def permute_batch(self, states: np.ndarray) -> np.ndarray:
        """[n][t][4] -> permuted copy."""
        out = np.ascontiguousarray(states, dtype=np.uint64).copy()
        n = out.size // (self.cfg.t * 4)
        _lib.check(_lib.lib().pmx_permute_batch(self._h, _ptr(out), n))
        return out

    def permute_batch_inplace(self, states: np.ndarray) -> None:
        """pmx_permute_batch on the caller's buffer itself (C-contiguous uint64 [n][t][4]; pinned_empty() memory
        takes the pipelined PCIe path)."""
        assert states.dtype == np.uint64 and states.flags["C_CONTIGUOUS"]
        _lib.check(_lib.lib().pmx_permute_batch(self._h, _ptr(states), states.size // (self.cfg.t * 4)))

    def hash_batch(self, msgs: np.ndarray, in_len: int, out_len: int, n: Optional[int] = None) -> np.ndarray:
        msgs = np.ascontiguousarray(msgs, dtype=np.uint64)
        if n is None:
            n = msgs.size // (in_len * 4)
        out = np.zeros((n, out_len, 4), dtype=np.uint64)
        _lib.check(_lib.lib().pmx_hash_batch(self._h, _ptr(msgs) if msgs.size else None, in_len, _ptr(out), out_len, n))
        return out

    def sponge_absorb_batch(self, states, tag, index, elems, in_len):
        n = tag.shape[0]
        _lib.check(_lib.lib().pmx_sponge_absorb_batch(self._h, _ptr(states), _ptr(tag), _ptr(index),
                                                      _ptr(elems) if elems.size else None, in_len, n))

    def sponge_squeeze_batch(self, states, tag, index, out_len) -> np.ndarray:
        n = tag.shape[0]
        out = np.zeros((n, out_len, 4), dtype=np.uint64)
        _lib.check(_lib.lib().pmx_sponge_squeeze_batch(self._h, _ptr(states), _ptr(tag), _ptr(index), _ptr(out),
                                                       out_len, n))
        return out

    def merkle_2to1(self, leaves: np.ndarray, want_nodes: bool = True):
        leaves = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, 4)
        m = leaves.shape[0]
        nodes = np.zeros((2 * m - 1, 4), dtype=np.uint64) if want_nodes else None
        root = np.zeros(4, dtype=np.uint64)
        _lib.check(_lib.lib().pmx_merkle_2to1(self._h, _ptr(leaves), m, _ptr(nodes), _ptr(root)))
        return nodes, root

    def merkle_2to1_forest(self, leaves: np.ndarray, n_trees: int, want_nodes: bool = True):
        """n_trees trees over leaves [n_trees][m][4], advanced together level by level (pmx_merkle_2to1_forest).  Returns
        (nodes, roots): nodes [n_trees * (2m - 1)][4] level-major - all leaves, then level 1 of every tree, ... - or None."""
        leaves = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, 4)
        m = leaves.shape[0] // n_trees
        assert m * n_trees == leaves.shape[0]
        nodes = np.zeros((n_trees * (2 * m - 1), 4), dtype=np.uint64) if want_nodes else None
        roots = np.zeros((n_trees, 4), dtype=np.uint64)
        _lib.check(_lib.lib().pmx_merkle_2to1_forest(self._h, _ptr(leaves), n_trees, m, _ptr(nodes), _ptr(roots)))
        return nodes, roots

    def merkle_2to1_forest_dev(self, d_nodes: int, n_trees: int, leaves_per_tree: int, stream: int = 0) -> None:
        _lib.check(_lib.lib().pmx_merkle_2to1_forest_dev(self._h, d_nodes, n_trees, leaves_per_tree, stream))

    # ---- device-pointer entry points (only enqueue; pointers are raw device addresses) ---------
    def permute_batch_dev(self, d_states: int, n: int, stream: int = 0) -> None:
        _lib.check(_lib.lib().pmx_permute_batch_dev(self._h, d_states, n, stream))

    def hash_batch_dev(self, d_in: int, in_len: int, d_out: int, out_len: int, n: int, stream: int = 0) -> None:
        _lib.check(_lib.lib().pmx_hash_batch_dev(self._h, d_in, in_len, d_out, out_len, n, stream))

    def sponge_absorb_batch_dev(self, d_states, d_tag, d_index, d_in, in_len, n, stream=0) -> None:
        _lib.check(_lib.lib().pmx_sponge_absorb_batch_dev(self._h, d_states, d_tag, d_index, d_in, in_len, n, stream))

    def sponge_squeeze_batch_dev(self, d_states, d_tag, d_index, d_out, out_len, n, stream=0) -> None:
        _lib.check(_lib.lib().pmx_sponge_squeeze_batch_dev(self._h, d_states, d_tag, d_index, d_out, out_len, n, stream))

    def merkle_2to1_dev(self, d_nodes: int, n_leaves: int, stream: int = 0) -> None:
        _lib.check(_lib.lib().pmx_merkle_2to1_dev(self._h, d_nodes, n_leaves, stream))


# ---------------------------------------------------------------------------------------------------
# Sponges
# ---------------------------------------------------------------------------------------------------
class BatchPoseidonSponge:
    """n independent PoseidonSponges advanced together: `state` [n][t][4], `mode_tag`/`mode_index` [n]."""

    def __init__(self, parameters: PoseidonConfig, n: int, device: int = 0):
        self.parameters = parameters
        self.n = n
        self.device = device
        self.state = np.zeros((n, parameters.t, 4), dtype=np.uint64)       # mod.rs:220
        self.mode_tag = np.full(n, MODE_ABSORBING, dtype=np.uint32)        # Absorbing{0}, mod.rs:221-223
        self.mode_index = np.zeros(n, dtype=np.uint32)

    @classmethod
    def new(cls, parameters: PoseidonConfig, n: int, device: int = 0) -> "BatchPoseidonSponge":
        return cls(parameters, n, device)

    def clone(self) -> "BatchPoseidonSponge":
        c = BatchPoseidonSponge(self.parameters, self.n, self.device)
        c.state, c.mode_tag, c.mode_index = self.state.copy(), self.mode_tag.copy(), self.mode_index.copy()
        return c

    def absorb(self, elems: np.ndarray) -> None:
        """elems [n][L][4]: every sponge absorbs its own L native field elements (mod.rs:232-254)."""
        elems = np.ascontiguousarray(elems, dtype=np.uint64)
        L = elems.size // (self.n * 4) if self.n else 0
        if L == 0:
            return                                                         # mod.rs:234-236
        self.parameters.context(self.device).sponge_absorb_batch(self.state, self.mode_tag, self.mode_index, elems, L)

    def squeeze_native_field_elements(self, num_elements: int) -> np.ndarray:
        """[n][num_elements][4]  (mod.rs:321-341)."""
        return self.parameters.context(self.device).sponge_squeeze_batch(self.state, self.mode_tag, self.mode_index,
                                                                        num_elements)

    # SpongeExt
    def into_state(self):
        return self.state, self.mode_tag, self.mode_index

    @classmethod
    def from_state(cls, state, parameters: PoseidonConfig, device: int = 0) -> "BatchPoseidonSponge":
        st, tag, idx = state
        sp = cls(parameters, st.shape[0], device)
        sp.state = np.ascontiguousarray(st, dtype=np.uint64).copy()
        sp.mode_tag = np.ascontiguousarray(tag, dtype=np.uint32).copy()
        sp.mode_index = np.ascontiguousarray(idx, dtype=np.uint32).copy()
        return sp


class PoseidonSponge:
    """One duplex sponge with the reference's public fields `parameters`, `state`, `mode`."""

    def __init__(self, parameters: PoseidonConfig, device: int = 0):
        self._b = BatchPoseidonSponge(parameters, 1, device)

    @classmethod
    def new(cls, parameters: PoseidonConfig, device: int = 0) -> "PoseidonSponge":
        return cls(parameters, device)

    @property
    def parameters(self) -> PoseidonConfig:
        return self._b.parameters

    @property
    def state(self) -> np.ndarray:                     # [t][4]
        return self._b.state[0]

    @property
    def mode(self) -> DuplexSpongeMode:
        return DuplexSpongeMode(int(self._b.mode_tag[0]), int(self._b.mode_index[0]))

    def clone(self) -> "PoseidonSponge":
        c = PoseidonSponge.__new__(PoseidonSponge)
        c._b = self._b.clone()
        return c

    def absorb(self, input) -> None:
        """CryptographicSponge::absorb (mod.rs:232-254).  `input` is either an `Absorb` object (absorb.py: its
        to_sponge_field_elements encoding is absorbed, as in the reference) or native field elements given
        directly as [L][4] Montgomery limbs."""
        from .absorb import Absorb
        if isinstance(input, Absorb):
            elems = self.parameters.field.from_ints(input.to_sponge_field_elements_as_vec(self.parameters.field))
        else:
            elems = input
        elems = np.ascontiguousarray(elems, dtype=np.uint64).reshape(1, -1, 4)
        self._b.absorb(elems)

    def fork(self, domain: bytes) -> "PoseidonSponge":
        """CryptographicSponge::fork (src/lib.rs:149-157): clone, then absorb len(domain) as usize bytes ++ domain,
        as a Vec<u8>."""
        from .absorb import Bytes, Usize
        new_sponge = self.clone()
        new_sponge.absorb(Bytes(Usize(len(domain)).to_sponge_bytes_as_vec() + bytes(domain)))
        return new_sponge

    def squeeze_native_field_elements(self, num_elements: int) -> np.ndarray:
        return self._b.squeeze_native_field_elements(num_elements)[0]

    def squeeze_field_elements(self, num_elements: int, field2: Optional[Field] = None):
        """squeeze_field_elements::<F2> (mod.rs:306-317).  Native field (default): identical to the native squeeze
        ([n][4] limbs).  Another field: the bit-recomposition of src/lib.rs:61-100, returned as canonical integers."""
        if field2 is None or field2.modulus == self.parameters.field.modulus:
            return self.squeeze_native_field_elements(num_elements)
        return self.squeeze_field_elements_with_sizes([None] * num_elements, field2)

    def squeeze_native_field_elements_with_sizes(self, sizes) -> List[int]:
        """FieldBasedCryptographicSponge::squeeze_native_field_elements_with_sizes (src/lib.rs:166-182): all sizes
        Full (None) -> the plain native squeeze; otherwise the bit-recomposition default with F = the native field.
        Returned as canonical integers."""
        f = self.parameters.field
        if all(sz is None for sz in sizes):
            return f.to_ints(self.squeeze_native_field_elements(len(sizes)))
        return self._squeeze_field_elements_with_sizes_default_impl(sizes, f)

    def squeeze_field_elements_with_sizes(self, sizes, field2: Field) -> List[int]:
        """PoseidonSponge's override (mod.rs:288-304): a field of the native characteristic goes through
        squeeze_native_field_elements_with_sizes (full native elements when every size is Full); any other field
        takes the bit-recomposition default.  `sizes`: None for Full, an int for Truncated(n).  Canonical integers."""
        if field2.modulus == self.parameters.field.modulus:
            return self.squeeze_native_field_elements_with_sizes(sizes)
        return self._squeeze_field_elements_with_sizes_default_impl(sizes, field2)

    def _squeeze_field_elements_with_sizes_default_impl(self, sizes, field2: Field) -> List[int]:
        """squeeze_field_elements_with_sizes_default_impl (src/lib.rs:61-100): every requested element takes
        MODULUS_BIT_SIZE(F2) - 1 bits (FieldElementSize::num_bits ignores the Truncated value, src/lib.rs:45-52),
        little-endian, reduced mod p2."""
        if len(sizes) == 0:
            return []
        nb = field2.modulus_bit_size - 1
        for sz in sizes:
            if sz is not None and sz > field2.modulus_bit_size:
                raise ValueError("num_bits is greater than the capacity of the field.")   # src/lib.rs:48
        bits = self.squeeze_bits(nb * len(sizes))
        out = []
        for k in range(len(sizes)):
            window = bits[k * nb:(k + 1) * nb]
            out.append(sum(1 << i for i, b in enumerate(window) if b) % field2.modulus)
        return out

    def squeeze_bytes(self, num_bytes: int) -> bytes:                     # mod.rs:256-270
        f = self.parameters.field
        usable = (f.modulus_bit_size - 1) // 8
        n = (num_bytes + usable - 1) // usable
        out = bytearray()
        for v in f.to_ints(self.squeeze_native_field_elements(n)):
            out += v.to_bytes(32, "little")[:usable]
        return bytes(out[:num_bytes])

    def squeeze_bits(self, num_bits: int) -> List[bool]:                  # mod.rs:272-286
        f = self.parameters.field
        usable = f.modulus_bit_size - 1
        n = (num_bits + usable - 1) // usable
        bits: List[bool] = []
        for v in f.to_ints(self.squeeze_native_field_elements(n)):
            bits.extend(bool((v >> k) & 1) for k in range(usable))
        return bits[:num_bits]

    def into_state(self):
        return self._b.state[0].copy(), self.mode

    @classmethod
    def from_state(cls, state, parameters: PoseidonConfig, device: int = 0) -> "PoseidonSponge":
        st, mode = state
        sp = cls(parameters, device)
        sp._b.state[0] = np.ascontiguousarray(st, dtype=np.uint64).reshape(parameters.t, 4)
        sp._b.mode_tag[0] = mode.tag
        sp._b.mode_index[0] = mode.index
        return sp
