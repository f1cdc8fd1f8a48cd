"""sponge_amd: MI355X-native batched Poseidon permutation + duplex-sponge driver behind the
CryptographicSponge / PoseidonSponge surface of arkworks-rs/sponge.  The product is the C ABI in
include/poseidon_mi355x.h (libposeidon_mi355x.so, hand-written gfx950 HIP kernels); this package is the
thin host-side mirror of the reference interface used by the tests and the benchmark."""
from ._lib import MODE_ABSORBING, MODE_SQUEEZING, PmxError, lib  # noqa: F401
from .field import BLS12_381_FR, BN254_FR, FIELDS, Field  # noqa: F401
from .poseidon import (BatchPoseidonSponge, Context, DuplexSpongeMode, PoseidonConfig, PoseidonSponge,  # noqa: F401
                       find_poseidon_ark_and_mds, get_default_poseidon_parameters, pinned_empty,
                       poseidon_config_from_lfsr)
from .merkle import MerkleTree, verify_paths  # noqa: F401,E402
