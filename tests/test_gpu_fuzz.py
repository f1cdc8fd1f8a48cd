"""A short run of tools/diag/fuzz_configs.py inside the GPU suite: seeded random configurations (exponents 0 ... 2^64 - 1, odd and zero
round counts, every rate / capacity split of t = 2 ... 12, the benchmarked fields and random primes of 225 ... 255 bits) through permute,
hash, tree and the duplex driver against the C port.  The GPU sessions of a round run it with hundreds of configurations
(tools/gpu_r05.sh: stage `fuzz`); the reference accepts every such config (PoseidonConfig::new asserts shapes only,
src/poseidon/mod.rs:187-213)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_fuzzed_configurations_agree_with_the_c_port(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diag", "fuzz_configs.py"), "40", str(seed)],
                       capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.strip().splitlines()[-12:])
    assert r.returncode == 0 and "40 configurations, 0 failing cases" in r.stdout, tail + "\n" + r.stderr[-2000:]
