"""BASELINE config C1 ("plumbing"): wall time of ONE sponge through the host-buffer entry points - new; absorb(3
elements); squeeze(3) as the reference KAT does (src/poseidon/mod.rs:376-399) - and of PoseidonSponge.new itself
(context from the library's cache), next to the C restatement on one host core."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sponge_amd as S  # noqa: E402
from oracle import cref  # noqa: E402
from oracle import poseidon_oracle as O  # noqa: E402

f = S.BLS12_381_FR
cfg = S.get_default_poseidon_parameters(f, 2, False)
msg = f.from_ints([0, 1, 2])
sp = S.PoseidonSponge.new(cfg)
sp.absorb(msg)
sp.squeeze_native_field_elements(3)          # context creation + first launches
N = 300
t0 = time.perf_counter()
for _ in range(N):
    fresh = S.PoseidonConfig(f, cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.mds, cfg.ark, cfg.rate, cfg.capacity)
    S.PoseidonSponge.new(fresh).parameters.context(0)
new_us = (time.perf_counter() - t0) / N * 1e6
t0 = time.perf_counter()
for _ in range(N):
    s = S.PoseidonSponge.new(cfg)
    s.absorb(msg)                              # 1 permutation (rate 2, 3 elements)
    s.squeeze_native_field_elements(3)         # 2 permutations
gpu_us = (time.perf_counter() - t0) / N * 1e6
cr = cref.CRef(O.make_config(O.BLS12_381_FR, 255, 2, 17, 8, 31))
st = np.zeros((3, 4), dtype=np.uint64)
t0 = time.perf_counter()
for _ in range(N):
    s_, m_, i_ = cr.sponge_absorb(st, 0, 0, msg)
    cr.sponge_squeeze(s_, m_, i_, 3)
cpu_us = (time.perf_counter() - t0) / N * 1e6
print(json.dumps({"sponge_new_with_fresh_equal_config_us": new_us, "gpu_absorb3_squeeze3_us": gpu_us,
                  "gpu_us_per_permutation": gpu_us / 3, "cpu_port_absorb3_squeeze3_us_incl_ctypes": cpu_us}))
