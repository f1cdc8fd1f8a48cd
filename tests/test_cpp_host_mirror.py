"""The C++ host mirror of the reference interface (sponge_amd/host/poseidon_sponge.hpp): its reference-style test
program is built against the C-ABI library and run - host-only parts on CPU, the sponge KAT on the GPU."""
import os
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")


def build():
    subprocess.check_call(["make", "-C", HERE, "all"], stdout=subprocess.DEVNULL)
    return os.path.join(HERE, "test_poseidon_sponge")


def test_cpp_mirror_host_only():
    out = subprocess.run([build(), "--host-only"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_mirror_reference_tests_on_gpu():
    out = subprocess.run([build()], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
