"""bench.py's own arithmetic, on the CPU: the count of matrix-core products a permutation issues (it prices them in `valu_issue`), the
price object itself and the gather predictions of the N > 1 forms.  The counts are checked against the counter pass of the t = 9 kernel
(profiles/r06: 3,862 per wave) and against the schedule's definition (pmx_permute.hpp: permute_hybrid)."""
import argparse
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")


def by_definition(t, rf, rp, k, hist_rows):
    """layer by layer: a row over n_in inputs is 2 n_in products"""
    total = rf * t * 2 * t                                   # one dense layer (or the windows' entry layer) per full round
    left = rp
    first = True
    while left > 0:
        kw = (rp - (-(-rp // k) - 1) * k) if first else k     # the first window is the short one
        first = False
        total += t * 2 * (t - 1 + k)                         # the window's layer: t rows over t - 1 carried lanes + K S-box outputs
        if hist_rows:
            for row in range(2, k):                           # the rows of x_3 .. x_K: `row` inputs each, present while row < kw
                if row < kw:
                    total += 2 * row
        left -= kw
    return total


@pytest.mark.parametrize("t,rf,rp", [(9, 8, 57), (3, 8, 31), (3, 8, 56), (4, 8, 56), (5, 8, 60), (6, 8, 57), (7, 8, 57), (8, 8, 57), (9, 8, 9), (9, 8, 1), (5, 2, 0)])
def test_products_per_permutation_follow_the_schedule(t, rf, rp):
    k = min(9, t)
    assert bench.matrix_products_per_permutation(t, rf, rp, k, t >= 4) == by_definition(t, rf, rp, k, t >= 4)


def test_the_t9_count_is_the_counter_passes():
    assert bench.matrix_products_per_permutation(9, 8, 57, 9, True) == 3862      # profiles/r06/z_*_pmc_c3_stalls.txt: 3,862 per wave


def test_products_are_priced_between_the_plain_floor_and_one():
    v = {"valu_instructions_per_permutation": 101507, "ns_per_instruction_and_simd": 2.3705, "floor_ns_per_instruction_and_simd": 1.7593, "frac": 1.7593 / 2.3705}
    bench.price_products(v, 3862, 2.2728e9)
    spread, burst = v["frac_products_priced"].values()
    assert v["frac"] < spread < burst < 1.0
    assert abs(spread - 0.824) < 0.002 and abs(burst - 0.880) < 0.002
    w = {"valu_instructions_per_permutation": None}
    bench.price_products(w, 100, 2.2e9)                                            # no count: the product count alone
    assert w["matrix_products_per_permutation"] == 100 and "frac_products_priced" not in w


@pytest.mark.parametrize("gather,chunks", [("final", 8), ("step", 8), ("root", 8), ("overlap", 8), ("overlap-root", 4), ("none", 8)])
def test_gather_predictions(gather, chunks):
    args = argparse.Namespace(gather=gather, gather_chunks=chunks)
    n, t, world = 1 << 21, 3, 8
    ms, text = bench.gather_model(args, world, n, t, False)
    one = 1e3 * n * t * 32 / 153e9
    if gather == "none":
        assert ms is None and text is None
    elif gather.startswith("overlap"):
        assert abs(ms - one / chunks) < 1e-9 and "behind piece i's kernel" in text and ("to rank 0" in text) == gather.endswith("root")
    else:
        assert abs(ms - one) < 1e-9 and (("ncclSend" in text) == (gather == "root"))
    assert bench.gather_model(args, 1, n, t, False) == (None, None) and bench.gather_model(args, world, n, t, True) == (None, None)
