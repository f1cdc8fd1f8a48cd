"""The parity suite once more on the SYSTEM's HIP runtime.

`libposeidon_mi355x.so` links /opt/rocm's libamdhip64.  In the pytest process that is not the runtime that ends up loaded:
several test modules import torch, torch bundles its own (older) HIP runtime under the same SONAME, and whichever is loaded
first serves the whole process - so every `-m gpu` test in this process exercises the kernels on torch's runtime, while a
Rust or C++ caller of the C ABI (the drop-in's real user) gets the system's.  The two have behaved
differently (round 4: the stream-ordered allocator, tests/test_gpu_sponge_passes.py::test_many_contexts_...).  This runs
the torch-free part of the suite - every kernel family, every engine, the drivers - in a child process that never imports
torch, and checks that it stayed that way."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_suite_without_torch_in_the_process():
    code = r'''
import sys
import pytest
rc = pytest.main(["-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k", "not many_contexts",
                  "tests/test_gpu_parity.py", "tests/test_gpu_sponge_passes.py"])
assert "torch" not in sys.modules, "a test module pulled torch in: this run was not on the system HIP runtime"
maps = open("/proc/self/maps").read()
hip = sorted({line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line})
print("HIP runtime mapped:", hip)
assert hip and all(p.startswith("/opt/rocm") for p in hip), hip
sys.exit(int(rc))
'''
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=2400)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, out[-4000:]
    assert "HIP runtime mapped:" in out and " passed" in out, out[-2000:]
