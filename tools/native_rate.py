"""C2's kernel rate WITHOUT torch in the process: the library then runs on the system's HIP runtime (/opt/rocm), the one a Rust or
C++ caller gets, instead of the older runtime torch bundles (bench.py imports torch, as its contract prescribes).  Device memory and
synchronisation through the C ABI's own helpers; wall clock around K back-to-back launches.  usage: python tools/native_rate.py [log2 n]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sponge_amd as S  # noqa: E402
from sponge_amd import _lib, synth  # noqa: E402

assert "torch" not in sys.modules
lib = _lib.lib()
for name, field, rate, rp, log2n in (("C2  bls12_381_fr t=3", S.BLS12_381_FR, 2, 31, 20), ("C3  bn254_fr t=9", S.BN254_FR, 8, 57, 18)):
    if len(sys.argv) > 1:
        log2n = int(sys.argv[1])
    cfg = S.poseidon_config_from_lfsr(field, rate, 5, 8, rp)
    ctx = cfg.context(0)
    n, t = 1 << log2n, rate + 1
    host = synth.random_elements(field, n * t, 0x5EED0002)
    d = ctypes.c_void_p()
    _lib.check(lib.pmx_device_alloc(0, ctypes.byref(d), host.nbytes))
    _lib.check(lib.pmx_device_upload(0, d, ctypes.c_void_p(host.ctypes.data), host.nbytes, None))
    _lib.check(lib.pmx_stream_synchronize(0, None))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:          # spin-up: the clock ramps over the first launches
        for _ in range(8):
            ctx.permute_batch_dev(d.value, n, 0)
        _lib.check(lib.pmx_stream_synchronize(0, None))
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.permute_batch_dev(d.value, n, 0)
    _lib.check(lib.pmx_stream_synchronize(0, None))
    dt = time.perf_counter() - t0
    maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
    print("%s, 2^%d states: %.4f ms per launch -> %.4g permutations/s   (HIP runtime: %s)" % (name, log2n, dt / K * 1e3, n * K / dt, ", ".join(maps)))
    lib.pmx_device_free(0, d)
