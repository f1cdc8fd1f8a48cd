"""The absorb / squeeze batch driver on wide states (t = 4..9): passes of a wave-uniform permutation kernel
(sponge_amd/csrc/pmx_sponge_plan.hpp, pmx_device.hip: sponge_first_kernel / permute_listed_kernel) instead of a per-lane state machine, so that the
driver runs on the permutation engine of its width - the matrix-core one at t = 7..9.

Reference semantics: absorb src/poseidon/mod.rs:232-254 + absorb_internal :121-150; squeeze_native_field_elements
:321-341 + squeeze_internal :153-182 (incl. the `!= rate` test of :175).  Every sponge is checked against the C
restatement run sponge by sponge; all calls go through the C ABI."""
import ctypes

import numpy as np
import pytest

import sponge_amd as S
from sponge_amd import _lib, synth

pytestmark = pytest.mark.gpu

PALLAS = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
P25519 = (1 << 255) - 19
# (field name, modulus, prime bits, rate, alpha, RF, RP, expect the matrix-core engine)
CASES = [
    ("bn254_fr", None, 254, 8, 5, 8, 57, True),          # BASELINE configs[2]'s width
    ("bls12_381_fr", None, 255, 8, 17, 8, 57, True),     # the generic-exponent build of the hybrid engines
    ("bls12_381_fr", None, 255, 7, 5, 8, 57, True),
    ("bls12_381_fr", None, 255, 6, 5, 8, 57, True),
    ("pallas_fp", PALLAS, 255, 8, 5, 8, 57, True),       # a third field ...
    ("p25519", P25519, 255, 8, 5, 8, 57, True),          # ... and one whose residues exceed 32 balanced bytes (top byte 127: stored as Y - p)
    ("bls12_381_fr", None, 255, 5, 5, 8, 57, True),      # t = 6, 5, 4: the matrix-core engines of the narrower hybrids
    ("bls12_381_fr", None, 255, 4, 5, 8, 56, True),
    ("bls12_381_fr", None, 255, 3, 3, 8, 56, True),
]


def _config(field_name, modulus, bits, rate, alpha, rf, rp):
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.FIELDS[field_name] if modulus is None else S.Field(field_name, modulus)
    p = f.modulus
    return f, S.poseidon_config_from_lfsr(f, rate, alpha, rf, rp), cref.CRef(O.make_config(p, bits, rate, alpha, rf, rp))


def _engine_info(cfg, op, n, length):
    info = _lib.PmxEngineInfo()
    _lib.check(_lib.lib().pmx_ctx_engine_info(cfg.context()._h, op, n, length, ctypes.byref(info)))
    return info


def _run_script(f, cfg, cr, rate, capacity, n, layout, seed):
    """absorbs and squeezes of every interesting length on n sponges in mixed modes; every sponge against the C restatement"""
    t, r = rate + capacity, rate
    rng = np.random.default_rng(seed)
    batch = S.BatchPoseidonSponge.new(cfg, n)
    batch.state = synth.random_elements(f, n * t, seed=7 + rate).reshape(n, t, 4)
    if layout == "random":
        batch.mode_tag = rng.integers(0, 2, n).astype(np.uint32)
        batch.mode_index = rng.integers(0, r + 1, n).astype(np.uint32)
    else:
        tag = np.zeros(n, dtype=np.uint32)
        idx = np.zeros(n, dtype=np.uint32)
        tag[256:512] = 1                       # workgroup 1: Squeezing{rate} (permutes first), workgroup 0: Absorbing{0} (does not)
        idx[256:512] = r
        tag[512:] = rng.integers(0, 2, n - 512)
        idx[512:] = rng.integers(0, r + 1, n - 512)
        batch.mode_tag, batch.mode_index = tag, idx
    ref = [(batch.state[i].copy(), int(batch.mode_tag[i]), int(batch.mode_index[i])) for i in range(n)]
    ops = [("absorb", r + 1), ("squeeze", r), ("squeeze", 0), ("absorb", r), ("squeeze", 1), ("absorb", 1), ("squeeze", 2 * r + 1),
           ("squeeze", r), ("absorb", 2 * r + 2), ("absorb", 0), ("squeeze", r - 1)]
    for step, (op, length) in enumerate(ops):
        if op == "absorb":
            elems = synth.random_elements(f, n * max(length, 1), seed=100 + step).reshape(n, max(length, 1), 4)[:, :length]
            batch.absorb(np.ascontiguousarray(elems))
            if length:
                ref = [cr.sponge_absorb(s, m, i, elems[j]) for j, (s, m, i) in enumerate(ref)]
        else:
            out = batch.squeeze_native_field_elements(length)
            nxt = []
            for j, (s, m, i) in enumerate(ref):
                s2, m2, i2, o = cr.sponge_squeeze(s, m, i, length)
                assert np.array_equal(out[j], o), (op, length, step, j)
                nxt.append((s2, m2, i2))
            ref = nxt
        want_state = np.stack([s for s, _, _ in ref])
        bad = np.nonzero((batch.state != want_state).any(axis=(1, 2)))[0]
        assert bad.size == 0, (op, length, step, bad[:8])
        assert [int(x) for x in batch.mode_tag] == [m for _, m, _ in ref], (op, length, step)
        assert [int(x) for x in batch.mode_index] == [i for _, _, i in ref], (op, length, step)


@pytest.mark.parametrize("layout", ["random", "blocks"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-t{c[3] + 1}-a{c[4]}")
def test_wide_driver_mixed_modes_vs_c_oracle(case, layout):
    """n sponges (more than two workgroups, the last one ragged) in different modes and positions advance together
    through absorbs and squeezes of every interesting length: longer than the rate, exactly the rate (the lazy
    permutation; on the squeeze side the `:175` case), one element, nothing (squeeze(0) of an absorbing sponge
    permutes), several rates.  `blocks`: whole workgroups in ONE mode, so that some workgroups skip a pass others take."""
    field_name, modulus, bits, rate, alpha, rf, rp, mfma = case
    f, cfg, cr = _config(field_name, modulus, bits, rate, alpha, rf, rp)
    t, r = rate + 1, rate
    n = 2 * 256 + 77
    info = _engine_info(cfg, _lib.OP_ABSORB, n, r + 3)
    assert b"passes" in info.engine and bool(info.mfma_dense) == mfma and info.launches == -(-(r + 3) // r), (info.engine, info.launches)
    assert _engine_info(cfg, _lib.OP_PERMUTE, n, 0).mfma_dense == int(mfma)
    _run_script(f, cfg, cr, rate, 1, n, layout, 1000 * rate + alpha)


@pytest.mark.parametrize("rate,capacity", [(7, 2), (3, 6), (8, 0), (3, 3), (2, 2)])
def test_wide_driver_other_rate_capacity_splits(rate, capacity):
    """PoseidonConfig::new takes any split of the width into rate and capacity (mod.rs:187-213; the default tables use
    capacity 1 only): the rate part starts at `capacity`, which is where the passes add, copy and count.  Constants of the
    width, another split, mixed modes, every sponge against the C restatement."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.BLS12_381_FR
    t = rate + capacity
    rp = 57 if t >= 6 else 56
    base = S.poseidon_config_from_lfsr(f, t - 1, 5, 8, rp)
    cfg = S.PoseidonConfig(f, 8, rp, 5, base.mds, base.ark, rate, capacity)
    ob = O.make_config(O.BLS12_381_FR, 255, t - 1, 5, 8, rp)
    cr = cref.CRef(O.PoseidonConfig(ob.p, 8, rp, 5, ob.ark, ob.mds, rate, capacity))
    n = 2 * 256 + 77
    assert b"passes" in _engine_info(cfg, _lib.OP_SQUEEZE, n, rate + 1).engine
    _run_script(f, cfg, cr, rate, capacity, n, "random", 77 * rate + capacity)


def test_wide_driver_device_resident_modes_out_of_range_are_clamped():
    """Device-resident mode words are not validated by the host (the host-buffer entry points are): an index above the rate
    behaves as the rate, exactly as in the per-lane kernels (pmx_device.hip: absorb_kernel).  Through the ABI's own device
    memory helpers (no torch: this file also runs in a torch-free child process, tests/test_gpu_system_runtime.py)."""
    f, cfg, cr = _config("bn254_fr", None, 254, 8, 5, 8, 57)
    lib = _lib.lib()
    n, t, r = 300, 9, 8
    st = synth.random_elements(f, n * t, seed=21).reshape(n, t, 4)
    tag = np.zeros(n, dtype=np.uint32)
    idx = np.full(n, r + 5, dtype=np.uint32)
    idx[::2] = r
    elems = synth.random_elements(f, n * 3, seed=22).reshape(n, 3, 4)

    def to_device(arr):
        p = ctypes.c_void_p()
        _lib.check(lib.pmx_device_alloc(0, ctypes.byref(p), arr.nbytes))
        _lib.check(lib.pmx_device_upload(0, p, ctypes.c_void_p(arr.ctypes.data), arr.nbytes, None))
        return p

    d_st, d_tag, d_idx, d_in = (to_device(np.ascontiguousarray(x)) for x in (st, tag, idx, elems))
    _lib.check(lib.pmx_stream_synchronize(0, None))
    cfg.context().sponge_absorb_batch_dev(d_st.value, d_tag.value, d_idx.value, d_in.value, 3, n, 0)
    got, got_tag, got_idx = np.zeros_like(st), np.zeros_like(tag), np.zeros_like(idx)
    for dst, src in ((got, d_st), (got_tag, d_tag), (got_idx, d_idx)):
        _lib.check(lib.pmx_device_download(0, ctypes.c_void_p(dst.ctypes.data), src, dst.nbytes, None))
    _lib.check(lib.pmx_stream_synchronize(0, None))
    for p in (d_st, d_tag, d_idx, d_in):
        lib.pmx_device_free(0, p)
    for j in range(n):
        s, m, i = cr.sponge_absorb(st[j], 0, r, elems[j])
        assert np.array_equal(got[j], s) and int(got_idx[j]) == i == 3 and int(got_tag[j]) == 0, j


def test_many_contexts_alive_and_calls_of_every_size_interleaved():
    """Regression test for a failure only the soak found (tools/soak.py): with a dozen contexts - a dozen streams - alive and
    calls of very different sizes following each other, the driver's pass lists, then taken from the stream-ordered
    allocator (hipMallocAsync / hipFreeAsync), came back from the pool while still in use: calls with two or more listed
    passes lost sponges (mode words never rewritten; the first one at the soak's tenth launch set) and one run ended in a
    memory fault.  Only on the SYSTEM's HIP runtime (/opt/rocm, what a Rust or C++ caller gets): a process that has imported
    torch runs on torch's bundled runtime, which did not show it - so this runs in a child process without torch.  The lists
    now live in a block the context keeps per caller stream.  The soak's own first 60 launch sets: sixteen configs in turn,
    sizes from 1 to 70001, lengths up to 3 rate + 2, random modes; every sponge (a sample above 300) of absorb + squeeze,
    a permutation batch and a hash batch against the C port."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "600", "60"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=900, cwd=root)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "soak ok: 60 launches-sets" in out, out[-3000:]


def test_a_call_longer_than_65536_rates_is_refused_not_launched():
    """The pass driver launches one kernel per permutation a sponge can need; a call that would need more than 65536 of them is
    refused with PMX_ERR_ARG before anything is touched (the header says so; splitting the call is equivalent)."""
    f, cfg, _ = _config("bn254_fr", None, 254, 8, 5, 8, 57)
    lib = _lib.lib()
    h = cfg.context()._h
    dummy = ctypes.c_void_p(4096)                                      # never dereferenced: the length check comes first
    assert lib.pmx_sponge_squeeze_batch_dev(h, dummy, dummy, dummy, dummy, 8 * 65536 + 1, 1, None) == _lib.PMX_ERR_ARG
    assert b"65536 rates" in lib.pmx_last_error()
    assert lib.pmx_sponge_absorb_batch_dev(h, dummy, dummy, dummy, dummy, 8 * 65536 + 9, 1, None) == _lib.PMX_ERR_ARG
    # the largest legal call shape is accepted by the check (n = 0: nothing is enqueued)
    assert lib.pmx_sponge_squeeze_batch_dev(h, None, None, None, None, 8 * 65536, 0, None) == _lib.PMX_OK


def test_host_calls_longer_than_65536_rates_are_cut_into_pieces_the_reference_takes_any_length():
    """The reference's absorb / squeeze take any length (src/poseidon/mod.rs:232-254, 321-341) and the drop-in type passes a whole
    input as one call with n = 1; the device call is limited to 65536 rates, so the host-buffer entry points cut a longer call into
    pieces (pmx_api.cpp: sponge_host).  t = 3 (rate 2, the reference's default shape): absorb(65536 * 2 + 1) on one sponge, then a
    squeeze whose LAST piece would be exactly one rate from a sponge standing inside its rate - the cut that would skip a permutation
    through the test of mod.rs:175 - and the same on three sponges in different modes (n > 1: the pieces are packed row by row)."""
    f, cfg, cr = _config("bls12_381_fr", None, 255, 2, 5, 8, 31)
    r, t = 2, 3
    long_in = 65536 * r + 1
    for n in (1, 3):
        batch = S.BatchPoseidonSponge.new(cfg, n)
        batch.state = synth.random_elements(f, n * t, seed=61 + n).reshape(n, t, 4)
        batch.mode_tag = np.array([0, 1, 1][:n], dtype=np.uint32)
        batch.mode_index = np.array([1, 1, 2][:n], dtype=np.uint32)
        ref = [(batch.state[i].copy(), int(batch.mode_tag[i]), int(batch.mode_index[i])) for i in range(n)]
        elems = synth.random_elements(f, n * long_in, seed=62 + n).reshape(n, long_in, 4)
        batch.absorb(elems)
        ref = [cr.sponge_absorb(s_, m, i, elems[j]) for j, (s_, m, i) in enumerate(ref)]
        for j, (s_, m, i) in enumerate(ref):
            assert np.array_equal(batch.state[j], s_) and (int(batch.mode_tag[j]), int(batch.mode_index[j])) == (m, i), ("absorb", n, j)
        # leave every sponge Squeezing inside its rate (index 1), then ask for 65536 rates + one rate
        out = batch.squeeze_native_field_elements(1)
        ref2 = []
        for j, (s_, m, i) in enumerate(ref):
            s2, m2, i2, o = cr.sponge_squeeze(s_, m, i, 1)
            assert np.array_equal(out[j], o)
            ref2.append((s2, m2, i2))
        assert all(int(x) == 1 for x in batch.mode_index) and all(int(x) == 1 for x in batch.mode_tag)
        long_out = 65536 * r + r
        out = batch.squeeze_native_field_elements(long_out)
        for j, (s_, m, i) in enumerate(ref2):
            s2, m2, i2, o = cr.sponge_squeeze(s_, m, i, long_out)
            assert np.array_equal(out[j], o), ("squeeze", n, j)
            assert np.array_equal(batch.state[j], s2) and (int(batch.mode_tag[j]), int(batch.mode_index[j])) == (m2, i2), ("squeeze", n, j)


@pytest.mark.parametrize("name", ["bls_t3_a5_8_31", "bls_t3_a17_8_31", "bls_t3_a257_8_13"])
def test_t3_calls_above_the_quad_range_run_on_the_matrix_core_engine_and_agree_with_the_quad_kernels(name):
    """t = 3 - alpha = 5 (BASELINE configs[1]'s shape), alpha = 17 (the reference's own rate-2 default and the config of its only
    permutation KAT, src/test.rs:15, src/poseidon/mod.rs:376-399), alpha = 257 (its weights table, src/test.rs:23-31): calls of at
    least 32769 units run on HybridEngine<3,alpha,mfma,windows of 3> (alpha = 5 specialised, the others on the generic S-box: "0") -
    permute, hash, compress, and absorb / squeeze as passes -, smaller ones on the quad kernels (pmx_device.hip: t3_mfma).
    pmx_ctx_engine_info says which; the SAME sponges through both sides of the threshold must agree limb for limb (the first
    32768 of them in a second call), and a sample of the large call - the first and the last workgroup in full, 1500 at
    random - is checked against the C restatement sponge by sponge, in mixed modes, through absorbs and squeezes that take
    zero, one, two and three permutations."""
    from gpu_helpers import c_oracle, product_config
    cfg, cr = product_config(name), c_oracle(name)
    f, t, r = cfg.field, 3, 2
    big, small = (1 << 17) + 333, 32768
    a_hyb, a_quad = {5: (b"5", b"5"), 17: (b"0", b"17")}.get(cfg.alpha, (b"0", b"0"))
    for op in (_lib.OP_PERMUTE, _lib.OP_HASH, _lib.OP_COMPRESS, _lib.OP_ABSORB, _lib.OP_SQUEEZE):
        lo = _engine_info(cfg, op, small, 4)
        assert lo.engine.startswith(b"QuadEngine<" + a_quad) and lo.mfma_dense == 0 and lo.partial_window == 0, lo.engine
        for n in (small + 1, big):
            hi = _engine_info(cfg, op, n, 4)
            assert hi.engine.startswith(b"HybridEngine<3," + a_hyb + b",mfma,windows of 3>") and hi.mfma_dense == 1 and hi.partial_window == 3, hi.engine
            assert (b"passes" in hi.engine) == (op in (_lib.OP_ABSORB, _lib.OP_SQUEEZE))
    rng = np.random.default_rng(33)
    state0 = synth.random_elements(f, big * t, seed=91).reshape(big, t, 4)
    tag0 = rng.integers(0, 2, big).astype(np.uint32)
    idx0 = rng.integers(0, r + 1, big).astype(np.uint32)
    sample = np.unique(np.concatenate([np.arange(256), np.arange(big - 333 - 256, big), rng.integers(0, big, 1500)]))

    def run(n):
        b = S.BatchPoseidonSponge.new(cfg, n)
        b.state, b.mode_tag, b.mode_index = state0[:n].copy(), tag0[:n].copy(), idx0[:n].copy()
        outs = []
        for step, (op, length) in enumerate([("absorb", 3), ("squeeze", 2), ("absorb", 5), ("squeeze", 3), ("squeeze", 0), ("absorb", 2), ("squeeze", 1)]):
            if op == "absorb":
                elems = synth.random_elements(f, big * length, seed=200 + step).reshape(big, length, 4)[:n]
                b.absorb(np.ascontiguousarray(elems))
                outs.append(None)
            else:
                outs.append(b.squeeze_native_field_elements(length))
        return b, outs

    b_big, o_big = run(big)
    b_small, o_small = run(small)
    assert np.array_equal(b_big.state[:small], b_small.state) and np.array_equal(b_big.mode_tag[:small], b_small.mode_tag)
    assert np.array_equal(b_big.mode_index[:small], b_small.mode_index)
    for a, b in zip(o_big, o_small):
        assert (a is None and b is None) or np.array_equal(a[:small], b)
    # ... and the sample against the C restatement
    ops = [("absorb", 3), ("squeeze", 2), ("absorb", 5), ("squeeze", 3), ("squeeze", 0), ("absorb", 2), ("squeeze", 1)]
    for j in sample:
        s, m, i = state0[j].copy(), int(tag0[j]), int(idx0[j])
        for step, (op, length) in enumerate(ops):
            if op == "absorb":
                s, m, i = cr.sponge_absorb(s, m, i, _t3_elems(f, big, length, step)[j])
            else:
                s, m, i, o = cr.sponge_squeeze(s, m, i, length)
                assert np.array_equal(o_big[step][j], o), (int(j), step)
        assert np.array_equal(b_big.state[j], s) and int(b_big.mode_tag[j]) == m and int(b_big.mode_index[j]) == i, int(j)
    # permute and hash across the threshold: the same rows through both engines
    st = synth.random_elements(f, big * t, seed=92).reshape(big, t, 4)
    pm = f.modulus
    edges = [[0, 0, 0], [pm - 1] * 3, [1, 0, 0], [0, 0, 1], [pm - 1, 0, 1], [(1 << 254) % pm, (1 << 253) % pm, pm - 2], [2, pm - 1, 0]]
    for k, e in enumerate(edges):            # the states the reference's arithmetic has edges at, inside the big launch (rows 0..6 are in `sample`)
        st[k] = f.from_ints(e).reshape(t, 4)
    ctx = cfg.context(0)
    p_big, p_small = ctx.permute_batch(st), ctx.permute_batch(st[:small])
    assert np.array_equal(p_big[:small], p_small)
    assert np.array_equal(p_big[sample], cr.permute_batch(np.ascontiguousarray(st[sample]), threads=0))
    msgs = synth.random_elements(f, big * 4, seed=93).reshape(big, 4, 4)
    h_big, h_small = ctx.hash_batch(msgs, 4, 2), ctx.hash_batch(msgs[:small], 4, 2)
    assert np.array_equal(h_big[:small], h_small)
    assert np.array_equal(h_big[sample], cr.hash_batch(np.ascontiguousarray(msgs[sample]), 4, 2, threads=0))


_T3_CACHE = {}


def _t3_elems(f, big, length, step):
    if step not in _T3_CACHE:
        _T3_CACHE[step] = synth.random_elements(f, big * length, seed=200 + step).reshape(big, length, 4)
    return _T3_CACHE[step]
