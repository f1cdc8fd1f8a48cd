"""Register budgets of the headline kernels, checked at compile time (hipcc cross-compiles without a GPU): the permutation kernels of
BASELINE configs[1] (t = 3) and configs[2] (t = 9) must keep their occupancy WITHOUT spilling.  A spill does not fail parity - it shows
up as HBM writes (round 5: an unused branch of the row finish cost the t = 3 kernel 32 bytes of scratch per lane at four waves per
SIMD and 1.34 x the algorithmic writes, found by the WRITE_SIZE counter pass) - so the budget is pinned here, where every round's CPU
suite sees it.  `make asm1` compiles ONE kernel (sponge_amd/csrc/Makefile, PMX_TU = 99) in seconds."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sponge_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
@pytest.mark.parametrize("t,alpha,max_vgprs,waves", [(3, 5, 128, 4), (3, 0, 128, 4), (9, 5, 256, 2)])
def test_permute_kernel_keeps_its_occupancy_without_scratch(t, alpha, max_vgprs, waves):
    subprocess.check_call(["make", "-C", CSRC, "asm1", f"T={t}", f"ALPHA={alpha}"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rpt = open(os.path.join(CSRC, "build", f"one_t{t}.rpt")).read()
    assert f"HybridEngineILi{t}ELi{alpha}EEE" in rpt
    get = lambda key: int(re.search(key + r": (\d+)", rpt).group(1))
    assert get(r"ScratchSize \[bytes/lane\]") == 0, rpt[-1500:]
    assert get("VGPRs") + get("AGPRs") <= max_vgprs and get(r"Occupancy \[waves/SIMD\]") >= waves, rpt[-1500:]
