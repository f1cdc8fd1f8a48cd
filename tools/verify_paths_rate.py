"""Time of pmx_merkle_verify_paths for k authentication paths of a given depth (synthetic siblings: the verdicts are all
False, the work is the same): host buffers (pageable and page-locked) and device-resident (pmx_merkle_verify_paths_dev).
usage: verify_paths_rate.py [log2 k = 15] [depth = 24]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sponge_amd as S  # noqa: E402
from sponge_amd import _lib, synth  # noqa: E402

k = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 15)
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 24
field = S.FIELDS["bls12_381_fr"]
cfg = S.poseidon_config_from_lfsr(field, 2, 5, 8, 31)
ctx = cfg.context(0)
lib = _lib.lib()
leaves = synth.random_elements(field, k, 1)
paths = synth.random_elements(field, k * depth, 2).reshape(k, depth, 4)
idx = (np.arange(k, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << depth)
root = synth.random_elements(field, 1, 3)[0]


def host_call(lv, ix, pa, ok):
    _lib.check(lib.pmx_merkle_verify_paths(ctx._h, ctypes.c_void_p(lv.ctypes.data), ctypes.c_void_p(ix.ctypes.data),
                                           ctypes.c_void_p(pa.ctypes.data), depth, k, ctypes.c_void_p(root.ctypes.data),
                                           ctypes.c_void_p(ok.ctypes.data)))


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


ok = np.zeros(k, dtype=np.uint8)
ms_pageable = timed(lambda: host_call(leaves, idx, paths, ok))
p_leaves, p_paths, p_idx = S.pinned_empty((k, 4)), S.pinned_empty((k, depth, 4)), S.pinned_empty((k,))
p_leaves[:], p_paths[:], p_idx[:] = leaves, paths, idx
ms_pinned = timed(lambda: host_call(p_leaves, p_idx, p_paths, ok))
dev = torch.device("cuda", 0)
d = {n: torch.from_numpy(a.view(np.int64).copy()).to(dev) for n, a in (("leaves", leaves), ("paths", paths), ("idx", idx), ("root", root))}
d_ok = torch.zeros(k, dtype=torch.uint8, device=dev)
d_work = torch.zeros((k, 12), dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream()
torch.cuda.synchronize()


def dev_call():
    _lib.check(lib.pmx_merkle_verify_paths_dev(ctx._h, d["leaves"].data_ptr(), d["idx"].data_ptr(), d["paths"].data_ptr(), depth, k,
                                               d["root"].data_ptr(), d_ok.data_ptr(), d_work.data_ptr(), stream.cuda_stream))
    torch.cuda.synchronize()


ms_dev = timed(dev_call)
mb = k * depth * 32 / 1e6
print(f"{k} paths of depth {depth} ({mb:.1f} MB of siblings, {k * depth} compressions):")
print(f"  host buffers, pageable     {ms_pageable:8.3f} ms")
print(f"  host buffers, page-locked  {ms_pinned:8.3f} ms")
print(f"  device-resident            {ms_dev:8.3f} ms   ({k * depth / ms_dev * 1e3:.3e} compressions/s, {ms_dev / depth * 1e3:.1f} us per level)")
