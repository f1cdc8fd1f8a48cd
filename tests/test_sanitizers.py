"""ASan + UBSan over the host-side code (kernel algorithm templates compiled for the host, table preparation,
Grain-LFSR generator).  GPU sanitizers are not available on the pool, so this is the sanitizer coverage."""
import os
import subprocess

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck")


def test_asan_ubsan_host_paths(tmp_path):
    exe = str(tmp_path / "sanitize_main")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-Wno-unknown-pragmas", "-I", os.path.join(HERE, "..", "..", "sponge_amd", "csrc"),
                           os.path.join(HERE, "sanitize_main.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0 and "sanitized ok" in out.stdout, out.stdout + out.stderr
