"""Device groups through the C ABI on whatever GPUs the box has (1 on the development box, N on a multi-GPU node):
sharded permutation of host and device-resident batches, the RCCL gather, the sharded Merkle reduction - all against
the C restatement - and the live communicator's own account of its ranks."""
import ctypes
import os

import numpy as np
import pytest
import torch

import sponge_amd as S
from sponge_amd import _lib, mgpu, synth

from gpu_helpers import c_oracle, product_config

pytestmark = pytest.mark.gpu

NAME = "bls_t3_a5_8_31"


def _group():
    cfg = product_config(NAME)
    return cfg, mgpu.DeviceGroup.single_process(cfg, _lib.lib().pmx_device_count())


def test_group_info_reports_the_live_communicator():
    cfg, g = _group()
    info = g.info()
    ndev = _lib.lib().pmx_device_count()
    assert info["world"] == info["n_local"] == info["comm_ranks"] == ndev
    assert info["first_rank"] == info["comm_first_rank"] == 0
    assert info["devices"] == list(range(ndev)) and info["width"] == 3
    assert info["rccl_version"] >= 20000
    assert g.stream(0) != 0 and g.stream(ndev) == 0
    g.close()


@pytest.mark.parametrize("n", [1, 1000, 100003, (1 << 18) + 17])
def test_host_batch_sharded_over_all_devices(n):
    cfg, g = _group()
    states = synth.random_elements(cfg.field, n * 3, seed=0x5EED0040 + n).reshape(n, 3, 4)
    got = g.permute_batch(states)
    assert np.array_equal(got, c_oracle(NAME).permute_batch(states, threads=0))
    # page-locked memory takes the pipelined path of every device
    pinned = S.pinned_empty((n, 3, 4))
    pinned[:] = states
    g.permute_batch_inplace(pinned)
    assert np.array_equal(pinned, got)
    g.close()


@pytest.mark.parametrize("n_total", [1 << 16, (1 << 16) + 5])
def test_device_resident_shards_and_rccl_gather(n_total):
    """With W ranks the first size gathers with ncclAllGather (equal shards) and the second with the grouped ncclBroadcast
    (ragged shards: n_total % W != 0).  On a ONE-GPU box both sizes divide by the world size, so both take the equal
    branch: the ragged branch is only reached from two GPUs up (test_ragged_gather_needs_two_gpus says so out loud)."""
    cfg, g = _group()
    world = g.world
    whole = synth.random_elements(cfg.field, n_total * 3, seed=0x5EED0041).reshape(n_total, 3, 4)
    shards, alls = [], []
    for l, dev in enumerate(g.devices):
        start, count = g.local_span(n_total, l)
        shards.append(torch.from_numpy(whole[start:start + count].view(np.int64).copy()).to(f"cuda:{dev}"))
        alls.append(torch.zeros((n_total, 3, 4), dtype=torch.int64, device=f"cuda:{dev}"))
    for dev in g.devices:                     # the group's streams do not wait for torch's: finish the uploads and fills first
        torch.cuda.synchronize(dev)
    g.permute_shards_dev([s.data_ptr() for s in shards], n_total)
    g.all_gather_dev([s.data_ptr() for s in shards], [a.data_ptr() for a in alls], n_total, 3)
    g.synchronize()
    want = c_oracle(NAME).permute_batch(whole, threads=0)
    for l in range(world):
        assert np.array_equal(alls[l].cpu().numpy().view(np.uint64), want), f"gathered copy on local device {l}"
    g.close()


@pytest.mark.parametrize("root,chunks", [(0, None), (-1, 4), (0, 8), (-2, 3)])
def test_gather_to_one_rank_and_the_last_step_piece_by_piece(root, chunks):
    """pmx_mgpu_gather_dev (chunks None) and pmx_mgpu_permute_gather_dev on real RCCL with every GPU of the box: grouped ncclSend / ncclRecv
    over xGMI from two GPUs up, the own-piece copies alone on a one-GPU box.  root -2 = the LAST rank; ragged shards from two GPUs up.
    Only the receivers' buffers are written; the reads below run on the group's own streams, which must have waited for the transfers."""
    cfg, g = _group()
    world = g.world
    root = world - 1 if root == -2 else root
    n_total = (1 << 15) * world + (1 if world > 1 else 0)
    whole = synth.random_elements(cfg.field, n_total * 3, seed=0x5EED0048).reshape(n_total, 3, 4)
    receivers = list(range(world)) if root < 0 else [root]
    shards, alls = [], []
    for l, dev in enumerate(g.devices):
        start, count = g.local_span(n_total, l)
        shards.append(torch.from_numpy(whole[start:start + count].view(np.int64).copy()).to(f"cuda:{dev}"))
        alls.append(torch.zeros((n_total, 3, 4), dtype=torch.int64, device=f"cuda:{dev}") if l in receivers else None)
    for dev in g.devices:
        torch.cuda.synchronize(dev)
    ptrs = [a.data_ptr() if a is not None else 0 for a in alls]
    if chunks is None:
        g.permute_shards_dev([s.data_ptr() for s in shards], n_total)
        g.gather_dev([s.data_ptr() for s in shards], ptrs, n_total, 3, root)
    else:
        g.permute_gather_dev([s.data_ptr() for s in shards], ptrs, n_total, root, chunks)
    g.synchronize()
    want = c_oracle(NAME).permute_batch(whole, threads=0)
    for l in receivers:
        assert np.array_equal(alls[l].cpu().numpy().view(np.uint64), want), f"copy on local device {l}"
    for l in range(world):
        start, count = g.local_span(n_total, l)
        assert np.array_equal(shards[l].cpu().numpy().view(np.uint64), want[start:start + count]), f"shard {l}"
    g.close()


def test_ragged_gather_needs_two_gpus():
    """The grouped-ncclBroadcast branch of pmx_mgpu_all_gather_dev (n_total % world != 0) cannot be reached with one rank.
    An explicit skip on a one-GPU box, not a silent pass through the equal-shard branch."""
    ndev = _lib.lib().pmx_device_count()
    if ndev < 2:
        pytest.skip("one GPU: every n_total divides by world = 1, the ragged ncclBroadcast branch needs >= 2 ranks")
    cfg, g = _group()
    n_total = (1 << 14) * ndev + 1
    whole = synth.random_elements(cfg.field, n_total * 3, seed=0x5EED0047).reshape(n_total, 3, 4)
    shards, alls = [], []
    for l, dev in enumerate(g.devices):
        start, count = g.local_span(n_total, l)
        shards.append(torch.from_numpy(whole[start:start + count].view(np.int64).copy()).to(f"cuda:{dev}"))
        alls.append(torch.zeros((n_total, 3, 4), dtype=torch.int64, device=f"cuda:{dev}"))
    for dev in g.devices:
        torch.cuda.synchronize(dev)
    g.permute_shards_dev([s.data_ptr() for s in shards], n_total)
    g.all_gather_dev([s.data_ptr() for s in shards], [a.data_ptr() for a in alls], n_total, 3)
    g.synchronize()
    want = c_oracle(NAME).permute_batch(whole, threads=0)
    for l in range(g.world):
        assert np.array_equal(alls[l].cpu().numpy().view(np.uint64), want), f"gathered copy on local device {l}"
    g.close()


@pytest.mark.parametrize("shape", ["equal", "ragged"])
def test_one_process_per_gpu_with_create_rank(shape, tmp_path):
    """The multi-process form (what bench.py --gpus N and a Rust job with one process per GPU use): one FRESH child
    process per visible GPU, no torch in them, the communicator id made by rank 0 (pmx_mgpu_unique_id) and handed over
    through a file, pmx_mgpu_create_rank (ncclCommInitRank) in every child.  Every rank checks the whole of its own
    gathered copy and the sharded Merkle root against the C restatement (tests/mgpu_rank_worker.py).  world =
    pmx_device_count(): one rank on the development box (which still runs the whole script: id hand-over, one-rank
    communicator, gather, tree), N ranks with RCCL over xGMI on a multi-GPU node."""
    import json
    import os
    import subprocess
    import sys
    world = _lib.lib().pmx_device_count()
    if shape == "ragged" and world < 2:
        pytest.skip("one GPU: no ragged split of the batch exists for world = 1")
    n_total = (1 << 14) * world + (3 if shape == "ragged" else 0)
    here = os.path.dirname(os.path.abspath(__file__))
    uid_file = str(tmp_path / "uid.bin")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(world):
        out = str(tmp_path / f"rank{r}.json")
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(here, "mgpu_rank_worker.py"), str(r), str(world), str(r),
                                             uid_file, str(n_total), "10", out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    results = []
    for out, p in procs:
        try:
            log, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for _, q in procs:
                q.kill()
            raise
        assert os.path.exists(out), log.decode(errors="replace")[-3000:]
        results.append(json.load(open(out)))
    for res in results:
        assert res["ok"], res
        assert res["info"]["comm_ranks"] == world and res["gather_bad_spans"] == []
        if world & (world - 1) == 0:
            assert res["tree_ok"] is True


def test_fan_out_carries_a_device_failure_back_to_the_caller():
    """pmx_mgpu_permute_batch / _hash_batch run one host thread per device; a failure on any device must come back to the
    calling thread with that device's message (the error text is thread-local), and a thread that cannot be started must
    not take the process down: its shard runs on the calling thread.  pmx_mgpu_test_fault injects both - a hook that exists only
    in libposeidon_mi355x_test.so (include/poseidon_mi355x_testing.h): when this process holds the shipped library, the test runs
    itself once more in a child pytest that binds the test-hook build (--pmx-test-library, tests/conftest.py)."""
    if not _lib.is_test_library():
        import subprocess
        import sys
        p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "--pmx-test-library", "-k",
                            "test_fan_out_carries_a_device_failure_back_to_the_caller"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
        assert p.returncode == 0 and b"1 passed" in p.stdout, p.stdout.decode(errors="replace")[-3000:]
        return
    cfg, g = _group()
    lib = _lib.lib()
    n = 3000
    states = synth.random_elements(cfg.field, n * 3, seed=0x5EED0048).reshape(n, 3, 4)
    want = c_oracle(NAME).permute_batch(states, threads=0)
    last = g.n_local - 1
    try:
        _lib.check(lib.pmx_mgpu_test_fault(last, 0))
        with pytest.raises(S.PmxError) as ei:
            g.permute_batch(states)
        assert ei.value.code == _lib.PMX_ERR_HIP and f"device {g.devices[last]} (slot {last}): injected failure" in str(ei.value)
        msgs = synth.random_elements(cfg.field, 100 * 4, seed=1).reshape(100, 4, 4)
        with pytest.raises(S.PmxError, match="injected failure"):
            g.hash_batch(msgs, 4, 1)
        _lib.check(lib.pmx_mgpu_test_fault(-1, 1))              # no worker threads: every shard on the calling thread
        assert np.array_equal(g.permute_batch(states), want)
        _lib.check(lib.pmx_mgpu_test_fault(0, 1))
        with pytest.raises(S.PmxError, match="injected failure"):
            g.permute_batch(states)
    finally:
        _lib.check(lib.pmx_mgpu_test_fault(-1, 0))
    assert np.array_equal(g.permute_batch(states), want)       # the group is intact afterwards
    g.close()


def test_hash_driver_per_shard_then_gather():
    """The absorb/squeeze batch driver sharded the same way: every rank hashes its rows with the single-device entry point
    on the group's context and stream, the digests are gathered over RCCL (row_elems = out_len)."""
    cfg, g = _group()
    n_total, in_len, out_len = 50000 + 3, 5, 2
    msgs = synth.random_elements(cfg.field, n_total * in_len, seed=0x5EED0044).reshape(n_total, in_len, 4)
    d_in, d_out, d_all = [], [], []
    for l, dev in enumerate(g.devices):
        start, count = g.local_span(n_total, l)
        d_in.append(torch.from_numpy(msgs[start:start + count].view(np.int64).copy()).to(f"cuda:{dev}"))
        d_out.append(torch.zeros((count, out_len, 4), dtype=torch.int64, device=f"cuda:{dev}"))
        d_all.append(torch.zeros((n_total, out_len, 4), dtype=torch.int64, device=f"cuda:{dev}"))
    for dev in g.devices:
        torch.cuda.synchronize(dev)
    for l in range(g.n_local):
        _, count = g.local_span(n_total, l)
        g.context(l).hash_batch_dev(d_in[l].data_ptr(), in_len, d_out[l].data_ptr(), out_len, count, g.stream(l))
    g.all_gather_dev([t.data_ptr() for t in d_out], [t.data_ptr() for t in d_all], n_total, out_len)
    g.synchronize()
    want = c_oracle(NAME).hash_batch(msgs, in_len, out_len, threads=0)
    assert np.array_equal(g.hash_batch(msgs, in_len, out_len), want)           # the host-batch form (pmx_mgpu_hash_batch)
    for l in range(g.n_local):
        assert np.array_equal(d_all[l].cpu().numpy().view(np.uint64), want)
    g.close()


@pytest.mark.parametrize("log2_leaves", [0, 1, 5, 16])
def test_sharded_merkle_root(log2_leaves):
    cfg, g = _group()
    m = 1 << log2_leaves
    if m < g.world:
        pytest.skip("fewer leaves than ranks")
    if g.world & (g.world - 1):
        pytest.skip("the sharded tree needs a power-of-two number of ranks")
    leaves = synth.random_elements(cfg.field, m, seed=0x5EED0042 + log2_leaves)
    assert np.array_equal(g.merkle_root(leaves), c_oracle(NAME).merkle(leaves, threads=0)[-1])
    g.close()


def test_one_rank_group_with_a_unique_id():
    """The multi-process constructor (what bench.py uses under torchrun), here as a world of one."""
    cfg = product_config(NAME)
    uid = mgpu.unique_id()
    assert len(uid) == 128 and any(uid)
    g = mgpu.DeviceGroup.one_rank(cfg, 0, 0, 1, uid)
    assert g.info()["comm_ranks"] == 1
    n = 5000
    states = synth.random_elements(cfg.field, n * 3, seed=0x5EED0043).reshape(n, 3, 4)
    d = torch.from_numpy(states.view(np.int64).copy()).to("cuda:0")
    out = torch.zeros_like(d)
    torch.cuda.synchronize()
    g.permute_shards_dev([d.data_ptr()], n)
    g.all_gather_dev([d.data_ptr()], [out.data_ptr()], n, 3)
    g.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), c_oracle(NAME).permute_batch(states, threads=0))
    g.close()


def test_bad_group_arguments():
    cfg = product_config(NAME)
    ndev = _lib.lib().pmx_device_count()
    with pytest.raises(S.PmxError) as ei:
        mgpu.DeviceGroup.single_process(cfg, ndev + 1)
    assert ei.value.code == _lib.PMX_ERR_ARG
    with pytest.raises(S.PmxError):
        mgpu.DeviceGroup.single_process(cfg, devices=[0, 0])
    _, g = _group()
    with pytest.raises(S.PmxError):
        g.merkle_root(np.zeros((3, 4), dtype=np.uint64))          # not a power of two
    g.close()


def test_shared_context_cache():
    """pmx_ctx_acquire: equal configs share one device context; idle contexts stay resident; the two ownership
    families do not mix."""
    from sponge_amd.poseidon import c_config
    lib = _lib.lib()
    a = S.poseidon_config_from_lfsr(S.BLS12_381_FR, 2, 5, 8, 31)
    b = S.poseidon_config_from_lfsr(S.BLS12_381_FR, 2, 5, 8, 31)      # separate object, same contents
    c = S.poseidon_config_from_lfsr(S.BLS12_381_FR, 2, 17, 8, 31)
    ha, hb, hc = a.context()._h.value, b.context()._h.value, c.context()._h.value
    assert ha == hb and ha != hc
    assert lib.pmx_ctx_destroy(ctypes.c_void_p(ha)) == _lib.PMX_ERR_ARG          # acquired contexts are released, not destroyed
    h = ctypes.c_void_p()
    cc = c_config(a)
    _lib.check(lib.pmx_ctx_acquire(ctypes.byref(cc), 0, ctypes.byref(h)))
    assert h.value == ha
    _lib.check(lib.pmx_ctx_release(h))
    own = ctypes.c_void_p()
    _lib.check(lib.pmx_ctx_create(ctypes.byref(cc), 0, ctypes.byref(own)))
    assert own.value != ha
    assert lib.pmx_ctx_release(own) == _lib.PMX_ERR_ARG
    _lib.check(lib.pmx_ctx_destroy(own))
    # a sponge per transcript: 200 PoseidonSponge.new on fresh-but-equal configs reuse the one context
    st = synth.random_elements(S.BLS12_381_FR, 2, seed=7)
    outs = set()
    for _ in range(200):
        cfg = S.PoseidonConfig(a.field, a.full_rounds, a.partial_rounds, a.alpha, a.mds, a.ark, a.rate, a.capacity)
        sp = S.PoseidonSponge.new(cfg)
        sp.absorb(st)
        outs.add(sp.squeeze_native_field_elements(1).tobytes())
        assert cfg.context()._h.value == ha
    assert len(outs) == 1
    # idle contexts can be dropped at any time; live ones stay usable and a dropped config is simply rebuilt
    d = S.poseidon_config_from_lfsr(S.BLS12_381_FR, 2, 257, 8, 13)
    d.context().close()                                          # idle now
    _lib.check(lib.pmx_ctx_cache_clear())
    d._ctx.clear()
    assert np.array_equal(d.context().permute_batch(st.reshape(1, 2, 4)[:, :1].repeat(3, axis=1)).shape, (1, 3, 4))
    assert a.context()._h.value == ha                            # still referenced: untouched by the clear


def test_shared_context_is_safe_across_threads():
    """Sponges made from equal configs share one cached device context; its host-buffer entry points serialise on the
    context's lock (ctypes drops the GIL during the calls, so these threads really overlap)."""
    import threading
    base = S.poseidon_config_from_lfsr(S.BLS12_381_FR, 2, 5, 8, 31)
    msgs = synth.random_elements(S.BLS12_381_FR, 8 * 5, seed=0x5EED0045).reshape(8, 5, 4)
    want = [c_oracle(NAME).hash_batch(msgs[i:i + 1], 5, 3, threads=1)[0] for i in range(8)]
    batch = synth.random_elements(S.BLS12_381_FR, 3000 * 3, seed=0x5EED0046).reshape(3000, 3, 4)
    want_batch = c_oracle(NAME).permute_batch(batch, threads=0)
    errors = []

    def worker(i):
        try:
            cfg = S.PoseidonConfig(base.field, base.full_rounds, base.partial_rounds, base.alpha, base.mds, base.ark, 2, 1)
            for _ in range(20):
                sp = S.PoseidonSponge.new(cfg)
                sp.absorb(msgs[i])
                if not np.array_equal(sp.squeeze_native_field_elements(3), want[i]):
                    errors.append(("sponge", i))
                if not np.array_equal(cfg.context().permute_batch(batch), want_batch):
                    errors.append(("batch", i))
        except Exception as e:                      # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]


def test_group_and_path_edge_cases():
    """Empty batches are no-ops, a tree of one leaf is its own root, a path of depth 0 compares the leaf with the root."""
    import ctypes
    cfg, g = _group()
    lib = _lib.lib()
    assert lib.pmx_mgpu_permute_batch(g._h, None, 0) == _lib.PMX_OK
    empty = (ctypes.c_void_p * g.n_local)()
    assert lib.pmx_mgpu_all_gather_dev(g._h, empty, empty, 0, 3) == _lib.PMX_OK
    assert lib.pmx_mgpu_permute_shards_dev(g._h, None, 5) == _lib.PMX_ERR_ARG
    one = synth.random_elements(cfg.field, 1, seed=5)
    if g.world == 1:
        assert np.array_equal(g.merkle_root(one), one[0])
    ok = S.verify_paths(cfg, one, [0], np.zeros((1, 0, 4), dtype=np.uint64), one[0])
    assert ok.tolist() == [True]
    assert S.verify_paths(cfg, one, [0], np.zeros((1, 0, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64)).tolist() == [False]
    assert S.verify_paths(cfg, np.zeros((0, 4), dtype=np.uint64), [], np.zeros((0, 3, 4), dtype=np.uint64), one[0]).tolist() == []
    g.close()


def test_wide_driver_dev_calls_from_several_threads_on_their_own_streams():
    """The *_dev absorb / squeeze calls of ONE context from four host threads, each on its own stream with its own sponges:
    the pass lists live in a block per caller stream (pmx_ctx::pass_blocks) and the calls serialise only while they enqueue,
    so the threads' work overlaps on the device and must not disturb each other.  BN254 Fr t = 9 (passes on the matrix-core
    engine), several listed passes per call; every thread's sponges against the C port."""
    import threading
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.BN254_FR
    cfg = S.poseidon_config_from_lfsr(f, 8, 5, 8, 57)
    cr = cref.CRef(O.make_config(O.BN254_FR, 254, 8, 5, 8, 57))
    ctx = cfg.context(0)
    n, t, L, K = 3000, 9, 19, 26
    errors = []

    def worker(k):
        try:
            rng = np.random.default_rng(100 + k)
            st = synth.random_elements(f, n * t, seed=500 + k).reshape(n, t, 4)
            tag = rng.integers(0, 2, n).astype(np.int32)
            idx = rng.integers(0, 9, n).astype(np.int32)
            msgs = synth.random_elements(f, n * L, seed=600 + k).reshape(n, L, 4)
            stream = torch.cuda.Stream(device="cuda:0")
            with torch.cuda.stream(stream):
                d_st = torch.from_numpy(st.view(np.int64).copy()).to("cuda:0")
                d_tag, d_idx = torch.from_numpy(tag).to("cuda:0"), torch.from_numpy(idx).to("cuda:0")
                d_in = torch.from_numpy(msgs.view(np.int64).copy()).to("cuda:0")
                d_out = torch.zeros((n, K, 4), dtype=torch.int64, device="cuda:0")
            stream.synchronize()
            for _ in range(3):          # absorb + squeeze three times over: the modes of one round feed the next
                ctx.sponge_absorb_batch_dev(d_st.data_ptr(), d_tag.data_ptr(), d_idx.data_ptr(), d_in.data_ptr(), L, n, stream.cuda_stream)
                ctx.sponge_squeeze_batch_dev(d_st.data_ptr(), d_tag.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), K, n, stream.cuda_stream)
            stream.synchronize()
            got_st, got_out = d_st.cpu().numpy().view(np.uint64), d_out.cpu().numpy().view(np.uint64)
            got_tag, got_idx = d_tag.cpu().numpy(), d_idx.cpu().numpy()
            for j in rng.choice(n, 150, replace=False):
                s, m, i = st[j], int(tag[j]), int(idx[j])
                for _ in range(3):
                    s, m, i = cr.sponge_absorb(s, m, i, msgs[j])
                    s, m, i, o = cr.sponge_squeeze(s, m, i, K)
                if not (np.array_equal(got_st[j], s) and np.array_equal(got_out[j], o) and (int(got_tag[j]), int(got_idx[j])) == (m, i)):
                    errors.append((k, int(j)))
        except Exception as e:          # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


def test_wide_driver_many_short_lived_streams_do_not_pile_up_scratch():
    """A caller that goes through many streams (one per request, say): the context keeps one pass-list block per stream, gives
    back the blocks of drained or destroyed streams once more than sixteen have been seen, and frees an outgrown block when its
    stream has drained.  Forty streams, growing batch sizes on each; results against the C port, and device memory does not
    grow by forty blocks."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.BN254_FR
    cfg = S.poseidon_config_from_lfsr(f, 8, 5, 8, 57)
    cr = cref.CRef(O.make_config(O.BN254_FR, 254, 8, 5, 8, 57))
    ctx = cfg.context(0)
    t, L, K = 9, 19, 10
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(40):
        n = 20000 + 5000 * (k % 5)
        stream = torch.cuda.Stream(device="cuda:0")
        msgs = synth.random_elements(f, n * L, seed=700 + k).reshape(n, L, 4)
        with torch.cuda.stream(stream):
            d_st = torch.zeros((n, t, 4), dtype=torch.int64, device="cuda:0")
            d_tag, d_idx = torch.zeros(n, dtype=torch.int32, device="cuda:0"), torch.zeros(n, dtype=torch.int32, device="cuda:0")
            d_in = torch.from_numpy(msgs.view(np.int64).copy()).to("cuda:0")
            d_out = torch.zeros((n, K, 4), dtype=torch.int64, device="cuda:0")
        stream.synchronize()
        ctx.sponge_absorb_batch_dev(d_st.data_ptr(), d_tag.data_ptr(), d_idx.data_ptr(), d_in.data_ptr(), L, n, stream.cuda_stream)
        ctx.sponge_squeeze_batch_dev(d_st.data_ptr(), d_tag.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), K, n, stream.cuda_stream)
        stream.synchronize()
        if k % 8 == 0:
            pick = np.arange(0, n, 97)
            assert np.array_equal(d_out.cpu().numpy().view(np.uint64)[pick], cr.hash_batch(np.ascontiguousarray(msgs[pick]), L, K, threads=0)), k
        del d_st, d_tag, d_idx, d_in, d_out, stream
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    grown = free0 - torch.cuda.mem_get_info()[0]
    assert grown < 17 * (2 * 40000 * 4 + 4096) + (64 << 20), grown        # at most ~sixteen blocks of the largest size (+ allocator slack)
