"""Where does a wave of the t = 9 window kernel spend its time?  Needs the diagnostic build of the library (tools/ab/build_variant.sh phases
"-DPMX_PHASE_TIMING" "3", bound in place of sponge_amd/libposeidon_mi355x.so by tools/ab/session_phases.sh): permute_hybrid / matrix_rows_mfma_w
add the s_memtime ticks of every phase of every wave to a device array.  Prints ticks per wave and permutation and the share of each phase.
usage: python tools/diag/phase_ticks.py [log2 n = 18] [width = 9]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sponge_amd as S  # noqa: E402
from sponge_amd import _lib, synth  # noqa: E402

PHASES = ["0 (the end of the permutation)", "1 S-boxes of the full rounds (rolled loop through the scratch)", "2 S-boxes of the windows + their operand cut",
          "3 history rows (table fetch, products, finish)", "4 layer: the two barriers of a stage, tile write, fetch of the next row's stage",
          "5 layer: tile reads + matrix-core products of a stage", "6 layer: lane swaps, row finish, scratch write",
          "7 layer: operand cut in front, scratch reads behind"]
lib = _lib.lib()
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
t = int(sys.argv[2]) if len(sys.argv) > 2 else 9
field = S.BN254_FR if t == 9 else S.BLS12_381_FR
cfg = S.poseidon_config_from_lfsr(field, t - 1, 5, 8, 57)
ctx = cfg.context(0)
n = 1 << log2n
host = synth.random_elements(field, n * t, 0x5EED0002)
d = ctypes.c_void_p()
_lib.check(lib.pmx_device_alloc(0, ctypes.byref(d), host.nbytes))
_lib.check(lib.pmx_device_upload(0, d, ctypes.c_void_p(host.ctypes.data), host.nbytes, None))
_lib.check(lib.pmx_stream_synchronize(0, None))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    for _ in range(8):
        ctx.permute_batch_dev(d.value, n, 0)
    _lib.check(lib.pmx_stream_synchronize(0, None))
ticks = (ctypes.c_ulonglong * 16)()
assert lib.pmx_diag_phase_ticks(ticks, 1) == 0
K = 20
t0 = time.perf_counter()
for _ in range(K):
    ctx.permute_batch_dev(d.value, n, 0)
_lib.check(lib.pmx_stream_synchronize(0, None))
dt = time.perf_counter() - t0
assert lib.pmx_diag_phase_ticks(ticks, 1) == 0
waves = n // 64 * K
tot = sum(ticks)
print("t = %d, 2^%d states, %d launches: %.4f ms per launch (instrumented build), %.0f ticks per wave and permutation" % (t, log2n, K, dt / K * 1e3, tot / waves))
for i, name in enumerate(PHASES):
    print("  %-70s %10.0f ticks  %5.1f %%" % (name, ticks[i] / waves, 100.0 * ticks[i] / tot))
