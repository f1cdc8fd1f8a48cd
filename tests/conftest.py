import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--pmx-test-library", action="store_true", default=False,
                     help="bind sponge_amd/libposeidon_mi355x_test.so (the shipped objects + the device-group test hooks of "
                          "include/poseidon_mi355x_testing.h) instead of the shipped library, for the whole session")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if config.getoption("--pmx-test-library"):
        from sponge_amd import _lib
        _lib.use_test_library()


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden
