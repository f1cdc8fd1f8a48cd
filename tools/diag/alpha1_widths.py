#!/usr/bin/env python3
"""Which (width, batch) of an alpha = 1 config differ from the C port, and where (diagnosis of the run-time-width engine)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
import numpy as np
import sponge_amd as S
from sponge_amd import _lib, synth
from oracle import cref, poseidon_oracle as O

f = S.BLS12_381_FR
for alpha in (1, 3):
    for rate in (2, 3, 4, 5, 6, 7, 8, 9, 11):
        t = rate + 1
        cfg = S.poseidon_config_from_lfsr(f, rate, alpha, 8, 57)
        cr = cref.CRef(O.make_config(O.BLS12_381_FR, 255, rate, alpha, 8, 57))
        info = _lib.PmxEngineInfo()
        for n in (1, 64, 333, 4096):
            _lib.check(_lib.lib().pmx_ctx_engine_info(cfg.context()._h, _lib.OP_PERMUTE, n, 0, ctypes.byref(info)))
            states = synth.random_elements(f, n * t, seed=7 * t + alpha).reshape(n, t, 4)
            got = cfg.context().permute_batch(states)
            want = cr.permute_batch(states, threads=0)
            bad = np.argwhere((got != want).any(axis=2))
            print("alpha %d t %2d n %5d engine %-40s mismatching (state, lane) pairs: %d of %d  first %s" % (
                alpha, t, n, info.engine.decode(), len(bad), n * t, bad[:6].tolist()), flush=True)
