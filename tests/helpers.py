"""Shared test helpers: the named configs of tests/golden/make_golden.py rebuilt through the oracle."""
import functools
import json
import os

from oracle import poseidon_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FIELDS = {"bls12_381_fr": (O.BLS12_381_FR, 255), "bn254_fr": (O.BN254_FR, 254)}


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def ints(xs):
    return [int(x, 16) for x in xs]


@functools.lru_cache(maxsize=None)
def oracle_config(name: str) -> O.PoseidonConfig:
    if name == "reference_test_a17_8_29":
        d = golden("reference_test_config.json")
        return O.PoseidonConfig(O.BLS12_381_FR, d["full_rounds"], d["partial_rounds"], d["alpha"],
                                [ints(r) for r in d["ark"]], [ints(r) for r in d["mds"]],
                                d["rate"], d["capacity"])
    d = golden("config_pins.json")[name]
    p, bits = FIELDS[d["field"]]
    return O.make_config(p, bits, d["rate"], d["alpha"], d["full_rounds"], d["partial_rounds"])


def prime_bits(name: str) -> int:
    return 254 if name.startswith("bn254") else 255
